"""gf_copy_list.hpp (the back end's one-kernel upload, the tracker's hand-overs) walked on the CPU: tests/native/copy_list_host.hip builds lists with the
library's own builder and runs copy_list_kernel's per-thread work for every (block, thread) of the launch on host memory -- every byte of every table moved
exactly once, nothing else touched, units as wide as the alignment allows, block shares contiguous.  hipcc compiles it (the header holds a __global__);
no HIP call is made, so it runs without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("flags", [[], ["-DGF_COPY_LIST_V2"]], ids=["default mapping", "division-free mapping (prepared, off by default)"])
def test_copy_lists_move_every_byte_once(tmp_path, flags):
    exe = tmp_path / "copy_list_host"
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O1", "-std=c++17"] + flags + ["-o", str(exe),
                           os.path.join(ROOT, "tests", "native", "copy_list_host.hip")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(": ok") >= 34
