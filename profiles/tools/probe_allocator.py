"""What the caching allocator of this torch build does with the workspaces of the side stream (profiles/r06_ab_notes.md section 4)."""
import os, torch
print("torch", torch.__version__, "alloc conf", os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"))
print("backend", torch.cuda.memory.get_allocator_backend())
dev = torch.device("cuda:0")
side = torch.cuda.Stream(priority=-1)
a = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
with torch.cuda.stream(side):
    b = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
    c = torch.empty(64 << 10, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for s in torch.cuda.memory_snapshot():
    print({k: (hex(v) if k == "address" else v) for k, v in s.items() if k != "blocks" and k != "frames"}, [(hex(b["address"]) if "address" in b else None, b["size"], b["state"]) for b in s["blocks"]][:4])
try:
    print(torch._C._cuda_getAllocatorBackend())
except Exception as e:
    print("n/a", e)
