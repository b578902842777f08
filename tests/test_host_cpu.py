"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol, the host
logic that needs no GPU (tap geometry, candidate counts, annotation handling, state-dict layout), the
"fail loudly without a GPU" rule, and the multi-process (gloo, world_size 2) pieces of the N > 1 path."""

from __future__ import annotations

import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def lib():
    from range_view_3d_detection_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "range_view_3d_detection_amd", "csrc"), "-j8"], check=True)
    return _lib


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_library_exports_every_declared_symbol(lib, tag):
    """Both builds of the C-ABI library (bf16 operands: training; fp16 operands: inference under autocast(float16)) load without
    a GPU and export every symbol include/rv3d.h declares."""
    handle = lib.load(tag)
    declared = lib.declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(handle, s)]
    assert not missing, missing
    assert handle.rv_version() >= 100
    assert handle.rv_pad_channels(5) == 32 and handle.rv_pad_channels(256) == 256
    with lib.operand(tag):
        assert lib.load() is handle and lib.act_dtype() == (torch.float16 if tag == "f16" else torch.bfloat16)
    assert lib.operand_tag() == "bf16"


def test_host_side_geometry(lib):
    h = lib.load()
    g = lib.TapGeom(3, 3, 1, 1, 1, 256, 256)
    assert h.rv_packed_weight_bytes(ctypes.byref(g)) == 9 * 256 * 256 * 2
    s = lib.TapShape(4, 64, 2048, 2048, 256, 256, lib.OUT_STATS)
    info = (ctypes.c_int32 * 4)()
    assert h.rv_tap_launch_info(ctypes.byref(g), ctypes.byref(s), 0, info) == 0
    assert list(info) == [6, 128, 64 * 4 * 4, 2]  # tapconv6: (16 rows x 32 cols) pixel tiles x two 128-channel tiles
    # statistic rows of a persistent launch: 4 wave rows per group of workgroups sharing a pixel tile (256 workgroups / 2 channel tiles)
    assert h.rv_tap_stats_rows(ctypes.byref(g), ctypes.byref(s), 0) == 4 * 128
    # kernel-selection hints travel PER CALL in rvTapShape.flags (the library holds no mutable option state)
    assert not hasattr(h, "rv_set_option")
    s5 = lib.TapShape(4, 64, 2048, 2048, 256, 256, lib.OUT_STATS | lib.SEL_NO_GEN6)
    assert h.rv_tap_launch_info(ctypes.byref(g), ctypes.byref(s5), 0, info) == 0
    assert list(info) == [5, 256, 64 * 8 * 4, 1]  # tapconv5<256>: (8 rows x 32 cols) pixel tiles x one 256-channel tile
    assert h.rv_tap_stats_rows(ctypes.byref(g), ctypes.byref(s5), 0) == 2 * 64 * 8 * 4
    # the same layer without generations 5 / 6, and a pointwise layer: tapconv4<256>, (4 rows x 64 cols) pixel tiles
    s4 = lib.TapShape(4, 64, 2048, 2048, 256, 256, lib.OUT_STATS | lib.SEL_NO_GEN5)
    assert h.rv_tap_launch_info(ctypes.byref(g), ctypes.byref(s4), 0, info) == 0 and list(info) == [4, 256, 32 * 16 * 4, 1]
    with lib.select(lib.SEL_NO_GEN5):  # (the host-side helper ORs the hints into every TapShape built inside the block)
        assert lib.TapShape(4, 64, 2048, 2048, 256, 256, 0).flags == lib.SEL_NO_GEN5
    assert lib.TapShape(4, 64, 2048, 2048, 256, 256, 0).flags == 0
    # a crop below one round of CUs stays on the register-staged kernels unless the call lifts the tile-count heuristic
    crop = lib.TapShape(1, 16, 64, 64, 256, 256, 0)
    assert h.rv_tap_launch_info(ctypes.byref(g), ctypes.byref(crop), 0, info) == 0 and info[0] < 4
    crop6 = lib.TapShape(1, 16, 64, 64, 256, 256, lib.SEL_SMALL_GRIDS | lib.SEL_SMALL_GRIDS6)
    assert h.rv_tap_launch_info(ctypes.byref(g), ctypes.byref(crop6), 0, info) == 0 and info[0] == 6
    g1 = lib.TapGeom(1, 1, 1, 0, 0, 256, 256)
    # a pointwise C -> C layer: the streaming GEMM (generation 7: 256 persistent workgroups = 256 statistic rows); pinned back: tapconv4<256>
    assert h.rv_tap_launch_info(ctypes.byref(g1), ctypes.byref(s), 0, info) == 0 and list(info) == [7, 256, 256, 1]
    assert h.rv_tap_stats_rows(ctypes.byref(g1), ctypes.byref(s), 0) == 256
    s1 = lib.TapShape(4, 64, 2048, 2048, 256, 256, lib.OUT_STATS | lib.SEL_NO_POINTWISE)
    assert h.rv_tap_launch_info(ctypes.byref(g1), ctypes.byref(s1), 0, info) == 0 and list(info) == [4, 256, 32 * 16 * 4, 1]
    # a folded BatchNorm on the way in needs the register-staged kernel
    s_aff = lib.TapShape(4, 64, 2048, 2048, 256, 256, lib.OUT_STATS | lib.IN_AFFINE | lib.IN_RELU)
    assert h.rv_tap_launch_info(ctypes.byref(g), ctypes.byref(s_aff), 0, info) == 0
    assert info[0] == 2  # tapconv2 (2 rows x 64 columns x 128 channels, the folded BatchNorm applied in its operand staging)
    # strided conv: Wv must be Wu * stride
    bad = lib.TapShape(4, 64, 1000, 2048, 256, 256, 0)
    g2 = lib.TapGeom(3, 3, 2, 1, 1, 128, 128)
    assert h.rv_tap_launch_info(ctypes.byref(g2), ctypes.byref(bad), 0, info) != 0
    assert b"stride_w" in h.rv_last_error()
    # transposed conv (3,8)/s4: four phases
    g3 = lib.TapGeom(3, 8, 4, 1, 2, 128, 256)
    s3 = lib.TapShape(4, 64, 512, 2048, 128, 256, 0)
    assert h.rv_tap_launch_info(ctypes.byref(g3), ctypes.byref(s3), 1, info) == 0 and list(info) == [6, 128, 4 * 16 * 4 * 4, 2]
    rates = (ctypes.c_int32 * 3)(8, 2, 1)
    assert h.rv_decode_num_candidates(64, 2048, 3, rates) == 64 * (256 + 1024 + 2048)  # 212 992
    assert h.rv_decode_num_candidates(64, 2048, 0, rates) == 64 * 2048
    assert h.rv_wnms_workspace_bytes(ctypes.c_int64(50000)) > 2 * 50000 * 782 * 8
    assert h.rv_bn_bwd_rows(ctypes.c_int64(4 * 64 * 2048)) == 1024


def test_weight_gradient_split_plans(lib):
    """rv_tap_wgrad_info (host only): kernel generation, split-K slabs and workgroups of the weight-gradient launch.  One round of
    workgroups as close to 256 as the tile count allows; the 48-tile 512 <-> 512 layer takes the BALANCED split (five regular slices per
    tile + sixteen remainder workgroups that each finish three tiles: a sixth slab, all 256 CUs busy)."""
    h = lib.load()
    info = (ctypes.c_int32 * 4)()
    s = lib.TapShape(4, 64, 2048, 2048, 0, 0, 0)

    def plan(cu, cv, k):
        g = lib.TapGeom(k, k, 1, k // 2, k // 2, cu, cv)
        assert h.rv_tap_wgrad_info(ctypes.byref(g), ctypes.byref(s), info) == 0
        nbytes = h.rv_tap_wgrad_workspace_bytes(ctypes.byref(g), ctypes.byref(s))
        assert nbytes == info[1] * k * k * cu * cv * 4  # one fp32 slab per split
        return list(info)[:3]

    assert plan(512, 512, 3) == [3, 6, 256]       # wgrad3, 5 + 1 slabs, 240 regular + 16 remainder workgroups
    assert plan(256, 256, 3) == [3, 21, 252]      # 12 tiles x 21 slices: already 98 % of the CUs, plain split
    assert plan(128, 128, 3) == [3, 85, 255]
    assert plan(128, 128, 1) == [3, 256, 256]     # 1x1, one tile: 8192 chunks of 64 pixels over 256 workgroups (never fewer than 32 per workgroup)


def test_null_arguments_fail_with_message(lib):
    h = lib.load()
    assert h.rv_ew_combine(ctypes.c_int64(10), 32, None, 32, None, None, None, 0, None, None, None, 32, 0, None) != 0
    assert b"null" in h.rv_last_error()
    with pytest.raises(lib.RvError):
        lib.call("rv_yaw_to_quat", None, ctypes.c_int64(4), ctypes.c_int64(1), None, None)


def test_modules_fail_loudly_without_gpu():
    from range_view_3d_detection_amd import _lib
    from range_view_3d_detection_amd.math.ops.coding import decode_range_view
    from range_view_3d_detection_amd.nn.blocks import BasicBlock

    m = BasicBlock(8, 8)
    with pytest.raises(_lib.RvError, match="no CPU fallback"):
        m(torch.randn(1, 8, 4, 32))
    with pytest.raises(_lib.RvError, match="no CPU fallback"):
        decode_range_view(torch.randn(1, 8, 4, 32), torch.randn(1, 3, 4, 32), True)


def test_state_dict_keys_and_init_match_reference(golden):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_model import build_tiny

    g = golden("tiny_model")
    backbone, head = build_tiny()
    ref = g.sub("sd")
    ours = {**{f"backbone.{k}": v for k, v in backbone.state_dict().items()}, **{f"head.{k}": v for k, v in head.state_dict().items()}}
    assert set(ours) == set(ref)
    for k, v in ref.items():
        assert tuple(ours[k].shape) == tuple(v.shape), k
    # DenseHead init (dense_head.py:62-72): N(0, 0.01) conv weights, focal prior bias on the classification head
    cls = head.classification_head["1"]["0"]
    assert abs(float(cls.blocks[-1][0].bias[0]) + math.log((1 - 0.01) / 0.01)) < 1e-6
    assert 0.005 < float(cls.blocks[0][0].weight.std()) < 0.02
    assert float(head.regression_head["1"]["0"].blocks[-1][0].bias.abs().max()) == 0.0


def test_annotation_table_to_cuboids(golden):
    from oracle import targets as otgt
    from range_view_3d_detection_amd.nn.heads.detection_head import annotations_to_cuboids

    ann = golden("tiny_model")["annotations"]
    ours = annotations_to_cuboids(ann)
    assert np.allclose(ours, otgt.annotations_to_cuboids(ann).numpy(), rtol=0, atol=1e-12)
    assert annotations_to_cuboids(np.zeros((0, 13))).shape == (0, 10)


def test_weight_permutation_for_metakernel_fusion_conv():
    """Reference channel order c*9+k (F.unfold) <-> engine order k*Cpad+c, and back for the gradient."""
    from range_view_3d_detection_amd import engine as E

    w = torch.nn.Parameter(torch.randn(16, 16 * 9, 1, 1))
    layer = E.TapLayer(w, 1, (0, 0), False, in_perm=(16, 9))
    tw = layer._torch_weight()
    assert tw.shape == (16, 9 * 32, 1, 1)
    assert torch.equal(tw[3, 5 * 32 + 7, 0, 0], w[3, 7 * 9 + 5, 0, 0])
    assert float(tw[:, 16:32].abs().max()) == 0.0
    assert torch.equal(layer.unpermute_grad(tw), w.detach())


_WORKER = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
from range_view_3d_detection_amd import engine as E
import bench
# (1) sweeps are sharded: every rank draws its own synthetic sweeps
b = bench.synthetic_batch(2, 8, 32, seed=1234 + rank, device="cpu", boxes_per_sweep=2)
digest = torch.tensor([float(b["features"].double().sum())], dtype=torch.float64)
gathered = [torch.zeros_like(digest) for _ in range(world)]
dist.all_gather(gathered, digest)
# (2) SyncBN statistics: local partial rows (+ this rank's element count: UNEVEN on purpose) -> global totals and count in
#     ONE collective, identical on every rank
g = torch.Generator().manual_seed(rank)
partial = torch.randn(5 + 128, 2, 32, generator=g)
E.COLLECTIVES.reset()
local = torch.zeros(2, 32)
glob = E.allreduce_partial_rows(partial, 5, 1000 + 7 * rank, local)   # flat (2C + 1): totals, then the global count
tot = glob[:64].view(2, 32)
local_ok = torch.allclose(local, partial[:5].sum(0), atol=1e-5)
# several layers in ONE collective (conv_bn_many / the grouped backward): same totals, one call
E.COLLECTIVES.reset()
p2 = torch.randn(3 + 128, 2, 8, generator=g)
views = E.allreduce_partial_rows_many([(partial, 5, 1000 + 7 * rank, None), (p2, 3, 10 + rank, None)])
group_ok = E.COLLECTIVES.calls == 1 and torch.allclose(views[0][:64].view(2, 32), tot) and float(views[1][16]) == sum(10 + r for r in range(world))
E.COLLECTIVES.reset()
glob = E.allreduce_partial_rows(partial, 5, 1000 + 7 * rank)
count_ok = float(glob.view(-1)[64]) == sum(1000 + 7 * r for r in range(world)) and E.COLLECTIVES.calls == 1 and E.COLLECTIVES.bytes == 65 * 4
ref = torch.stack([torch.randn(5 + 128, 2, 32, generator=torch.Generator().manual_seed(r))[:5].sum(0) for r in range(world)]).sum(0)
# (2b) which BatchNorm holders are synchronised (engine.SYNC_BN = None: decided per layer)
bn, sbn = torch.nn.BatchNorm2d(4), torch.nn.SyncBatchNorm(4)
rules = [E.bn_sync_world(sbn, True) == world, E.bn_sync_world(sbn, False) == 1]
try:
    E.bn_sync_world(bn, True)
    rules.append(False)
except RuntimeError as e:
    rules.append("sync_batchnorm" in str(e))
E.SYNC_BN = True
rules.append(E.bn_sync_world(bn, True) == world)
E.SYNC_BN = False
rules.append(E.bn_sync_world(sbn, True) == 1 and E.bn_sync_world(bn, True) == 1)
E.SYNC_BN = None
# (2c) gradient averaging without DistributedDataParallel: flat buffer, one all-reduce per finished node
torch.manual_seed(0)
pa, pb, pc = (torch.nn.Parameter(torch.zeros(s)) for s in ((3, 5), (7,), (2, 2, 2)))
gs = E.GradSync([pa, pb, pc], world)
ga, gb, gc = (torch.full_like(p, float(rank + 1) * (i + 1)) for i, p in enumerate((pa, pb, pc)))
node1 = gs.reduce_node([pc], [gc])            # the node that finishes first (the towers) ...
node2 = gs.reduce_node([pa, pb], [ga, None])  # ... then the rest; pb got no gradient
taken = node1 == [None] and node2 == [None, None]  # (the views become p.grad in finish(), not through autograd)
gs.finish()
mean = sum(r + 1 for r in range(world)) / world
grad_ok = (taken and torch.allclose(pa.grad, torch.full((3, 5), mean * 1)) and pb.grad is None and torch.allclose(pc.grad, torch.full((2, 2, 2), mean * 3))
           and pa.grad.data_ptr() == gs.view(pa).data_ptr() and not gs.works)
# a node whose parameters are NOT contiguous in parameter order (a head with two strides / tasks: all classification towers are
# registered before the regression towers): [pa, pc] first, then [pb] -- every gradient averaged exactly once (round-3 ADVICE)
for p_ in (pa, pb, pc):
    p_.grad = None
gs.reduce_node([pa, pc], [ga, gc])
split_ok = len(gs.works) == 2  # two runs of the flat buffer, pb's region (stale) not among them
gs.reduce_node([pb], [gb])
gs.finish()
split_ok = (split_ok and torch.allclose(pa.grad, torch.full((3, 5), mean * 1)) and torch.allclose(pb.grad, torch.full((7,), mean * 2))
            and torch.allclose(pc.grad, torch.full((2, 2, 2), mean * 3)))
try:  # a second backward without zero_grad(set_to_none=True): the installed views would be accumulated into themselves
    gs.reduce_node([pa], [ga])
    split_ok = False
except Exception as e:
    split_ok = split_ok and "set_to_none" in str(e)
grad_ok = grad_ok and split_ok
# (3) step time = MAX over ranks (bench.py contract)
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = {"rank": rank, "digests": [float(x) for x in gathered], "stats_err": float((tot - ref).abs().max()), "tmax": float(t),
       "count_ok": bool(count_ok and local_ok and group_ok and grad_ok), "rules": rules}
print("RESULT " + json.dumps(out), flush=True)
dist.destroy_process_group()
"""


def test_two_process_gloo_sharding_and_syncbn_reduction(tmp_path, lib):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    import json

    res = []
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        res.append(json.loads([ln for ln in o.splitlines() if ln.startswith("RESULT ")][0][7:]))
    assert res[0]["digests"] == res[1]["digests"] and res[0]["digests"][0] != res[0]["digests"][1]
    assert all(r["stats_err"] < 1e-4 and r["tmax"] == 2.0 for r in res)
    assert all(r["count_ok"] and all(r["rules"]) for r in res), res


def test_detections_table_and_feather_files(tmp_path):
    """``build_dataframe`` / per-sweep feather files: schema, join rules and file layout."""
    import pyarrow as pa
    import pyarrow.feather as feather
    import torch

    from range_view_3d_detection_amd.math.ops.coding import build_dataframe, write_detections

    g = torch.Generator().manual_seed(0)
    params = torch.randn(7, 10, generator=g)
    scores = torch.rand(7, generator=g)
    categories = torch.tensor([0.0, 2.0, 1.0, 2.0, 0.0, 1.0, 2.0])  # floats, as the decoder returns them
    batch_index = torch.tensor([0.0, 0.0, 1.0, 1.0, 1.0, 0.0, 5.0])  # batch 5 has no uuid -> dropped by the inner join
    uuids = {"batch_index": [0, 1], "log_id": ["logA", "logB"], "timestamp_ns": [315969904359876000, 315969904459876000]}
    t = build_dataframe(params, scores, categories, batch_index, uuids, ["REGULAR_VEHICLE", "PEDESTRIAN", "BUS"])
    assert t.column_names == ["tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz", "score", "batch_index",
                              "log_id", "timestamp_ns", "category"]
    assert [str(f.type) for f in t.schema] == ["float"] * 11 + ["int32", "string", "int64", "string"]
    assert t.num_rows == 6
    assert t.column("category").to_pylist() == ["REGULAR_VEHICLE", "BUS", "PEDESTRIAN", "BUS", "REGULAR_VEHICLE", "PEDESTRIAN"]
    assert t.column("log_id").to_pylist() == ["logA", "logA", "logB", "logB", "logB", "logA"]
    assert torch.equal(torch.tensor(t.column("length_m").to_pylist()), params[:6, 3])
    paths = write_detections(t, str(tmp_path), "run0")
    assert [p.split("predictions/")[1] for p in paths] == ["run0/logA/315969904359876000.feather", "run0/logB/315969904459876000.feather"]
    a = feather.read_table(paths[0])
    assert a.schema == t.schema and a.num_rows == 3 and a.column("score").to_pylist() == [t.column("score")[i].as_py() for i in (0, 1, 5)]


def test_detections_table_matches_the_reference_build_dataframe(golden):
    """f2 pinned: ``build_dataframe`` (math/ops/coding.py:31-76) was run by the reference itself in the build container
    (``tests/golden/make_golden.py::gen_detections_frame``, over the polars stand-in) on the decoder rows of
    ``nms_wrapper.npz``; column order, dtypes and every row of this package's Arrow table must equal that frame."""
    import numpy as np
    import torch

    from range_view_3d_detection_amd.math.ops.coding import build_dataframe

    g, nw = golden("detections_frame"), golden("nms_wrapper")
    uuids = {"batch_index": g.np("uuids/batch_index").tolist(), "log_id": [str(x) for x in g.np("uuids/log_id")],
             "timestamp_ns": g.np("uuids/timestamp_ns").tolist()}
    t = build_dataframe(nw["b/tiny/params"], nw["b/tiny/scores"], nw["b/tiny/categories"], nw["b/tiny/batch_index"], uuids,
                        [str(x) for x in g.np("category_names")])
    cols = [str(c) for c in g.np("columns")]
    assert t.column_names == cols
    arrow = {"float32": "float", "int32": "int32", "int64": "int64"}
    for c, dt in zip(cols, [str(d) for d in g.np("dtypes")]):
        want = g.np(f"col/{c}")
        if dt.startswith("<U") or dt.startswith("|S") or dt == "object":
            assert str(t.schema.field(c).type) == "string" and t.column(c).to_pylist() == [str(x) for x in want], c
        else:
            assert str(t.schema.field(c).type) == arrow[dt], (c, dt)
            assert np.array_equal(np.asarray(t.column(c)), want), c
    assert 0 < t.num_rows < nw["b/tiny/params"].shape[0]  # sweep 1 has no uuid row: dropped by the inner join


def test_wnms_gpu_shim_importable_and_refuses_cpu_tensors():
    """The reference's ``import weighted_nms_ext`` resolves to compat/weighted_nms_ext.py; there is no CPU fallback."""
    import sys

    import torch

    from range_view_3d_detection_amd import compat

    sys.path.insert(0, list(compat.__path__)[0])
    try:
        import weighted_nms_ext
    finally:
        sys.path.pop(0)
    b = torch.zeros(4, 5)
    d = torch.zeros(4, 9)
    with pytest.raises(RuntimeError, match="GPU"):
        weighted_nms_ext.wnms_gpu(b, d, torch.zeros_like(d), torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long), 0.3, 0.5, 0)


def test_training_recipe_glue():
    """AdamW(1e-3) + OneCycleLR(max_lr = 0.00075 * sqrt(devices * batch)) stepped per optimisation step (nn/meta/arch.py:48-75)."""
    import math

    import torch

    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers, one_cycle_max_lr

    p = [torch.nn.Parameter(torch.zeros(3))]
    opt, sched = configure_optimizers(p, num_devices=8, batch_size=4, total_steps=50)
    assert isinstance(opt, torch.optim.AdamW) and isinstance(sched, torch.optim.lr_scheduler.OneCycleLR)
    assert abs(one_cycle_max_lr(0.00075, 8, 4) - 0.00075 * math.sqrt(32)) < 1e-12 and one_cycle_max_lr(1e-3, 8, 4, False) == 1e-3
    lrs = []
    for _ in range(50):
        opt.step()
        lrs.append(opt.param_groups[0]["lr"])
        sched.step() if len(lrs) < 50 else None
    assert abs(max(lrs) - 0.00075 * math.sqrt(32)) < 1e-6 and lrs[0] < lrs[10] and lrs[-1] < lrs[20]
    assert configure_optimizers(p, 1, 4, 10, debug=True)[1] is None


def test_bench_gpus_flag_launches_the_ranks_itself():
    """``python bench.py --gpus N`` is ONE command for N ranks (reference: scripts/train.sh:16-21 + conf/trainer/train.yaml:39-44):
    without WORLD_SIZE the process is a launcher that starts torch.distributed.run as a child BEFORE torch is imported (so nothing
    in it can have initialised HIP) and hands the child's exit code back; with a WORLD_SIZE that disagrees it refuses.  On this
    GPU-less box every rank stops at "needs an MI355X", which is how the test sees that two ranks were started."""
    if torch.cuda.device_count() > 0:
        pytest.skip("the GPU form of this test is tests/test_gpu_ddp.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--widths", "c32", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "[bench launcher] 2 ranks" in out.stderr and "torch imported in the launcher: False" in out.stderr
    assert "--nproc-per-node=2" in out.stderr and "--master-addr 127.0.0.1" in out.stderr
    assert out.stderr.count("bench.py needs an MI355X") == 2, out.stderr[-2000:]
    # a rank count that disagrees with the environment is refused (a record labelled N GPUs must come from N ranks)
    bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE="1", RANK="0"), cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "WORLD_SIZE=1" in bad.stderr
    # N = 1 is not a launcher
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert "[bench launcher]" not in one.stderr and "needs an MI355X" in one.stderr


def _gfx950_code_objects(path):
    """The gfx950 code objects of a HIP shared library (the clang offload bundles of its .hip_fatbin section)."""
    import re
    import struct
    import tempfile

    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, path], check=True)
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        hdr = data[m.start():]
        (num,) = struct.unpack_from("<Q", hdr, 24)
        off = 32
        for _ in range(num):
            o, sz, tl = struct.unpack_from("<QQQ", hdr, off)
            off += 24
            triple = hdr[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz > 0:
                out.append(hdr[o:o + sz])
    return out


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_wgrad3_epilogue_waits_before_it_touches_a_register(tmp_path, lib, tag):
    """Round 6's rv-waymo fault: wgrad3's transposing LDS reads are inline asm, the last one of the K loop is a prefetch nobody consumes, and the
    compiler -- which takes its destination registers for dead -- had scheduled the epilogue's slab-address arithmetic ABOVE the ``s_waitcnt`` that
    follows the loop, into two of those registers; a late LDS return then overwrote the address (stores to address 0).  The registers are now operands
    of the wait.  This checks the machine code of the library as built: in every instance of the kernel the ``s_waitcnt vmcnt(0) lgkmcnt(0)`` that
    closes the K loop is the FIRST instruction of the loop's exit block (what precedes it is the branch), not something behind vector arithmetic."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    path = lib.LIB_PATH_F16 if tag == "f16" else lib.LIB_PATH
    cos = [c for c in _gfx950_code_objects(path) if b"wgrad3_kernel" in c]
    assert len(cos) == 1
    co = tmp_path / "wgrad.co"
    co.write_bytes(cos[0])
    text = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(co)], check=True, capture_output=True, text=True).stdout
    start = next(i for i, ln in enumerate(text.splitlines()) if "wgrad3_kernel" in ln and ln.rstrip().endswith(">:"))
    lines = text.splitlines()[start + 1:]
    end = next((i for i, ln in enumerate(lines) if ln.rstrip().endswith(">:")), len(lines))
    body = [ln.split("//")[0].strip() for ln in lines[:end] if ln.strip()]
    waits = [i for i, ln in enumerate(body) if ln.startswith("s_waitcnt vmcnt(0) lgkmcnt(0)")]
    assert len(waits) == 3, waits  # one per tap-group size (1, 2, 3 taps)
    for i in waits:
        assert body[i - 1].startswith(("s_cbranch", "s_branch")), body[i - 6:i + 1]
        # ... and the first vector instruction behind the wait is arithmetic on registers nobody is still loading into
        nxt = next(ln for ln in body[i + 1:] if ln.startswith(("v_", "global_", "ds_")))
        assert nxt.startswith("v_"), nxt


def _kernel_body(text, name):
    lines = text.splitlines()
    start = next(i for i, ln in enumerate(lines) if name in ln and ln.rstrip().endswith(">:"))
    rest = lines[start + 1:]
    end = next((i for i, ln in enumerate(rest) if ln.rstrip().endswith(">:")), len(rest))
    return [ln.split("//")[0].strip() for ln in rest[:end] if ln.strip()]


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_pointwise_kernel_counts_its_wait_over_the_order_it_was_written_in(tmp_path, lib, tag):
    """``pointwise_kernel`` (csrc/posconv.hip) waits for the NEXT tile's eight LDS-DMA loads with ``s_waitcnt vmcnt(8)`` while this tile's eight
    16-byte stores stay in flight: the counter retires in issue order, so the wait is right only if the machine code issues the eight loads BEFORE
    the eight stores -- an order the compiler is free to change (the two touch different memory).  Checked on the library as built."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    path = lib.LIB_PATH_F16 if tag == "f16" else lib.LIB_PATH
    cos = [c for c in _gfx950_code_objects(path) if b"pointwise_kernel" in c]
    assert len(cos) == 1
    co = tmp_path / "posconv.co"
    co.write_bytes(cos[0])
    text = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(co)], check=True, capture_output=True, text=True).stdout
    body = _kernel_body(text, "pointwise_kernel")
    waits = [i for i, ln in enumerate(body) if ln.startswith("s_waitcnt vmcnt(8)")]
    assert len(waits) == 1, waits
    mem = [ln.split()[0] for ln in body[:waits[0]] if ln.startswith(("global_load_lds", "global_store", "global_load", "buffer_"))]
    assert mem[-16:] == ["global_load_lds_dwordx4"] * 8 + ["global_store_dwordx4"] * 8, mem[-20:]
