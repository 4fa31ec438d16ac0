"""Print VGPR / scratch usage per kernel from a hipcc -save-temps .s file (CPU container: no GPU needed).
usage: python tools/vgprs.py recstudio_amd/csrc/rsa_fused-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
names = re.findall(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', txt)
# metadata order differs between versions; parse kernel blocks instead
blocks = txt.split('- .agpr_count:')[1:]
for b in blocks:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    vg = re.search(r'\.vgpr_count:\s+(\d+)', b).group(1)
    ag = b.split('\n')[0].strip()
    sc = re.search(r'\.private_segment_fixed_size:\s+(\d+)', b).group(1)
    lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', b).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace('rsa::', '').replace('(rsa::FwdParams)', '')
    if flt in dem:
        print(f'vgpr={vg:>3s} agpr={ag:>3s} scratch={sc:>4s} lds={lds:>6s}  {dem[:110]}')
