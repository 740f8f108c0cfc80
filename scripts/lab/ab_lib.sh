#!/bin/bash
# Same-box A/B of two BUILDS of the library (e.g. a kernel variant kept as ase_amd/csrc/libase_hip_<tag>.so beside the product's
# libase_hip.so): every build twice, interleaved, on (1) the carrying NT shapes in f16 (back-to-back launches, HIP events) and
# (2) the benchmark update.  The build is chosen by patching ase_amd.lib.LIB_PATH in the driver process - the product has no
# environment switch for it.
#   bash scripts/lab/ab_lib.sh libase_hip.so libase_hip_eb.so [more builds ...] [precision]      (AB_EXTRA='--no-multi-stream': more bench.py arguments)
cd "$(dirname "$0")/../.."
LIBS=(); P=f16gpx3
for a in "$@"; do case "$a" in *.so) LIBS+=("$a");; *) P="$a";; esac; done      # any number of builds, then (optionally) the precision
for rep in $(seq 1 ${REPS:-2}); do
  for lib in "${LIBS[@]}"; do
    python - "$lib" "$P" 2>/dev/null <<'PY'
import sys, os, json, io, contextlib, runpy
sys.path.insert(0, os.getcwd())
import ase_amd.lib as L
L.LIB_PATH = os.path.join(os.getcwd(), 'ase_amd', 'csrc', sys.argv[1])
import torch
from ase_amd.backend import HipBackend
be = HipBackend()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = []
for M, N, K in [(16384, 1024, 1024), (32768, 1024, 1024), (32768, 1024, 320), (16384, 1024, 512), (12288, 1024, 1408), (131072, 1024, 1024),
                (32768, 1024, 512), (32768, 512, 256), (32768, 512, 64), (16384, 1024, 320), (12288, 1024, 512),
                (32768, 64, 1024), (32768, 64, 512), (16384, 64, 512), (32768, 64, 256), (12288, 128, 512)]:
    A = (torch.randn(M, K, device='cuda') * 0.5).half(); B = (torch.randn(N, K, device='cuda') * 0.1).half()
    C = torch.zeros(M, N, device='cuda', dtype=torch.float16); bias = torch.randn(N, device='cuda')
    bits = torch.zeros(M, N // 32, dtype=torch.int32, device='cuda')
    ms = timeit(lambda: be.gemm_nt(A, B, C, M, N, K, bias=bias, act=L.ACT_RELU, mask_out=bits))
    out.append(f'{M}x{N}x{K} {ms * 1e3:.1f}us {2 * M * N * K / ms / 1e9:.0f}TF')
print(sys.argv[1], 'NT f16:', ' | '.join(out), flush=True)
sys.argv = ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '3', '--precision', sys.argv[2], '--no-cpu-baseline', '--no-config5',
            '--throughput-mode', 'none', '--detail', ''] + os.environ.get('AB_EXTRA', '').split()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):          # (stderr stays a real file: bench.py enables faulthandler on it)
    runpy.run_path('bench.py', run_name='__main__')
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(os.path.basename(L.LIB_PATH), 'update:', d['ms_per_step'], 'ms', d['value'], 'samples/s; nt8 avg', d['roofline']['avg_launch_us'], 'us', flush=True)
PY
  done
done
