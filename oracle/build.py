"""Build recipe for the C part of the oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

``python -m oracle.build`` compiles ``oracle/c/oracle.c`` with gcc into
``oracle/_build/liboracle.so``.  ``-ffp-contract=off`` keeps fp32 arithmetic unfused so
the HIP kernels (compiled with the same flag) can be required to match bit for bit.

There is no ``oracle/_ref``: the reference is pure Python (zero native sources, SURVEY.md
§0), so there is nothing of the reference's to compile; its Python was run in the build
container to generate ``tests/golden/*.npz`` instead (``tests/golden/make_golden.py``).
"""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-o", OUT, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
