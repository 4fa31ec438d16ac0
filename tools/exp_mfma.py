import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libexp_mfma.so'))
dev = torch.device('cuda', 0)
out = torch.zeros(4, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
iters = 2000
for chains in (1, 2):
    for waves_per_simd in (1, 2, 3, 4):
        blocks = 256 * waves_per_simd          # 4 waves per block -> one per SIMD
        for _ in range(2):
            lib.exp_mfma(chains, blocks, iters, ctypes.c_void_p(out.data_ptr()), st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            lib.exp_mfma(chains, blocks, iters, ctypes.c_void_p(out.data_ptr()), st)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        flops = blocks * 4 * iters * 64 * 4096.0
        print(f'chains={chains} waves/SIMD={waves_per_simd}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s', flush=True)

src = torch.rand(64 * 64 + 1024, device=dev)
for waves_per_simd in (1, 2):
    blocks = 256 * waves_per_simd
    for _ in range(2):
        lib.exp_mfma_regs(blocks, iters, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(src.data_ptr()), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        lib.exp_mfma_regs(blocks, iters, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(src.data_ptr()), st)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f'64 B regs / 4 A regs, waves/SIMD={waves_per_simd}: {ms:.3f} ms  {blocks * 4 * iters * 64 * 4096.0 / ms / 1e9:.1f} TFLOP/s', flush=True)
