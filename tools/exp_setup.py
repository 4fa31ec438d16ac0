"""Set-up costs: PopularSamplerModel construction (pop_prob, table, bucket lines) at N = 1e7 / 1e8, on the device."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recstudio_amd as ra
from bench import zipf_counts
dev = torch.device('cuda', 0)
for N in (10_000_001, 100_000_001):
    t0 = time.perf_counter(); counts = zipf_counts(N, 100_000_000); t_counts = time.perf_counter() - t0
    for where in ('cpu->gpu', 'gpu'):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if where == 'gpu':
            ps = ra.PopularSamplerModel(counts.to(dev))
        else:
            ps = ra.PopularSamplerModel(counts).to(dev)
        kw = ps.lookup_kwargs()
        torch.cuda.synchronize(); t1 = time.perf_counter() - t0
        print(f'N={N}: synthetic counts {t_counts:.2f} s (bench helper, CPU); PopularSamplerModel [{where}] {t1:.2f} s; '
              f'lines 2^{kw.get("lines_log2")} = {kw["cdf_lines"].numel() * 4 / 1e6:.0f} MB', flush=True)
        del ps, kw
        torch.cuda.empty_cache()
