"""The streaming kernels are normally launched with one item per lane; their grid-stride loops (used only beyond
2^31 workgroups) are exercised here by re-running the element-wise parity tests in a child process whose launches
are capped at 61 workgroups (MPCX_MAX_GRID), so every kernel loops many times with a ragged tail."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parity_with_capped_grids():
    env = dict(os.environ, MPCX_MAX_GRID="61")
    sel = "relu_chain or msb_parity or private_compare or vecmul or misc_segments or small_segments or share_convert"
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider",
           os.path.join(HERE, "test_gpu_aby3.py"), os.path.join(HERE, "test_gpu_securenn.py"),
           os.path.join(HERE, "test_gpu_my.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert " passed" in res.stdout and "failed" not in res.stdout, tail
