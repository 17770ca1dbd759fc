"""Does a captured hipMemsetAsync node keep its order against neighbouring kernel nodes when a second process shares
the GPU?  Each process replays [memset(buf) ; buf = max(buf, small) ; chk = amax(buf) ; buf.fill_(1e30)] and reads chk
after every replay: anything but amax(small) means the kernel after the memset saw the scribble of the previous replay.

    python tools/probe_memset_node.py [n_procs] [replays]
"""
import ctypes
import subprocess
import sys

import torch


def worker(tag, replays, after_kernel=False):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    n = 16 * 128 * 128 * 80
    buf = torch.empty(n, device=dev)
    small = torch.rand(n, device=dev)
    want = float(small.amax())
    chk = torch.zeros((), device=dev)
    work = torch.randn(2048, 2048, device=dev)
    s = torch.cuda.Stream()
    bad = 0
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        buf.fill_(1e30)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            if after_kernel:          # the memset node then depends on a kernel node, as in the training step (preprocess first)
                work.mul_(1.0001)
            e = hip.hipMemsetAsync(buf.data_ptr(), 0, n * 4, torch.cuda.current_stream().cuda_stream)
            assert e == 0, e
            torch.maximum(buf, small, out=buf)
            chk.copy_(buf.amax())
            buf.fill_(1e30)
            work.mul_(1.0001).add_(1e-3)
        for i in range(replays):
            g.replay()
            v = float(chk)
            if v != want:
                bad += 1
                print(f"[{tag}] replay {i}: amax {v!r} (want {want!r})", flush=True)
    print(f"[{tag}] memset {'after a kernel node' if after_kernel else 'as root node'}: {replays} replays, {bad} with a stale buffer",
          flush=True)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        bad = worker(sys.argv[2], int(sys.argv[3])) + worker(sys.argv[2], int(sys.argv[3]), after_kernel=True)
        sys.exit(1 if bad else 0)
    procs_n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    procs = [subprocess.Popen([sys.executable, __file__, "--worker", f"p{i}", str(replays)]) for i in range(procs_n)]
    codes = [p.wait() for p in procs]
    print("exit codes", codes)
    sys.exit(max(codes))
