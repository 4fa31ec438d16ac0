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
json.dump(res, open(os.path.join(dst, f'{ROUND}_requests.json'), 'w'), indent=1, sort_keys=True)
open(os.path.join(dst, f'{ROUND}_requests.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
