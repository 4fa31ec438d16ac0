#!/bin/bash
# GPU box: A/B a -D switch of the full-score GEMM.  usage: exp_fs.sh MACRO v1 v2 ...
M=$1; shift
for v in "$@"; do
  rm -f recstudio_amd/csrc/rsa_fullscore.o
  make -C recstudio_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off -D$M=$v" > /dev/null 2>&1
  echo "== $M=$v: $(python - <<'PY'
import torch, time
import recstudio_amd as ra
dev = torch.device('cuda', 0)
torch.manual_seed(0)
item = torch.randn(1_000_001, 128, device=dev) * 0.02
q = torch.randn(2048, 128, device=dev) * 0.02
def t(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
a = t(lambda: ra.ops.fullscore(item, q, want_lse=True))
b = t(lambda: ra.ops.fullscore(item, q, want_lse=True, k=100))
c = t(lambda: ra.ops.fullscore(item, q, want_scores=True), 4)
q512 = q[:512].contiguous()
d = t(lambda: ra.ops.fullscore(item, q512, want_lse=True, k=100))
print(f'lse {a:.3f} ms ({2*2048*128*1e6/a/1e9:.1f} TF)  lse+top100 {b:.3f} ms  scores {c:.3f} ms  B=512 lse+top100 {d:.3f} ms')
PY
)"
done
rm -f recstudio_amd/csrc/rsa_fullscore.o
make -C recstudio_amd/csrc -j8 > /dev/null 2>&1
