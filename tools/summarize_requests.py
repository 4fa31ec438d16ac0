"""rocpd databases of tools/collect_requests.sh -> <round>_requests.{json,txt}: per shape, for the kernel with the most
fabric read requests, the averages per launch of every collected counter (launches after the first three)."""
import glob
import json
import os
import sqlite3
import sys

out_dir, dst, shapes = sys.argv[1], sys.argv[2], sys.argv[3].split()
ROUND = os.environ.get('ROUND', 'r04')
res, lines = {}, []
for shape in shapes:
    per = {}
    for dbp in glob.glob(os.path.join(out_dir, shape, 'g*', '**', '*.db'), recursive=True):
        c = sqlite3.connect(dbp)
        for name, counter, value in c.execute('select kernel_name, counter_name, value from counters_collection order by id'):
            if 'rsa::' in name:
                per.setdefault(name, {}).setdefault(counter, []).append(value)
    if not per:
        continue
    dom = max(per, key=lambda k: sum(per[k].get('TCC_EA0_RDREQ_sum', [0])))
    entry = {'kernel': dom.replace('void ', '')[:110]}
    for counter, vals in sorted(per[dom].items()):
        tail = vals[3:] if len(vals) > 4 else vals
        entry[counter] = round(sum(tail) / len(tail), 1)
    res[shape] = entry
    lines.append(f'## {shape}   {entry["kernel"]}')
    for k, v in entry.items():
        if k != 'kernel':
            lines.append(f'{k:32s} {v:16.1f} per launch')
# request rates over the kernel durations of the same collection (<round>_kernel_profiles.json next to the output, when there)
head = ['# tools/collect_requests.sh: L2 <-> fabric request counters of the dominant kernel, average per launch (launches after the',
        '# first three), one rocprofv3 --kernel-trace --pmc pass per counter group.  Reads leave the L2 as 128-byte requests only',
        '# (RDREQ_32B = 0: RDREQ x 128 B reproduces the corrected FETCH_SIZE), writes as 64-byte requests only (WRREQ_64B = WRREQ).']
kp = os.path.join(dst, f'{ROUND}_kernel_profiles.json')
if os.path.exists(kp):
    prof = json.load(open(kp))
    for shape, e in res.items():
        if shape in prof and 'TCC_EA0_RDREQ_sum' in e:
            us = prof[shape]['avg_us']
            e['kernel_avg_us'] = us
            e['read_requests_G_per_s'] = round(e['TCC_EA0_RDREQ_sum'] / us / 1e3, 1)
            e['read_write_requests_G_per_s'] = round((e['TCC_EA0_RDREQ_sum'] + e.get('TCC_EA0_WRREQ_sum', 0)) / us / 1e3, 1)
            head.append(f'# {shape}: {us} us per launch -> {e["read_requests_G_per_s"]} G read requests/s, '
                        f'{e["read_write_requests_G_per_s"]} G requests/s reads + writes')
lines = head + lines
json.dump(res, open(os.path.join(dst, f'{ROUND}_requests.json'), 'w'), indent=1, sort_keys=True)
open(os.path.join(dst, f'{ROUND}_requests.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
