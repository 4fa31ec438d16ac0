"""rocpd databases of one shape (tools/collect_profiles.sh) -> an entry of <round>_kernel_profiles.json and a block of
<round>_kernel_profiles.txt (round tag: $ROUND, default r04); the databases (tens of MB each) are deleted afterwards.
usage: python tools/summarize_shapes.py OUT_DIR DST_DIR SHAPE
Entry: dominant kernel (largest time per step among the rsa:: / rocprim / rccl kernels), its rocprofv3 average over the
launches after the warm-up, HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; the factor 2 is the gfx950
correction of MI355X_MICROARCH.md, section HBM; WRITE_SIZE as is), and for multi-kernel steps the per-step sums."""
import glob
import json
import os
import shutil
import sqlite3
import sys

out_dir, dst, shape = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def db(sub):
    hits = glob.glob(os.path.join(out_dir, shape, sub, '**', '*.db'), recursive=True)
    return sqlite3.connect(hits[0]) if hits else None


def steps_of(kind):
    try:
        for line in open(os.path.join(out_dir, f'{shape}.{kind}.log')):
            if line.startswith('{"shape"'):
                j = json.loads(line)
                return j['steps'], j['warmup']
    except OSError:
        pass
    return None, None


def ours(name):
    # (the placement probe runs while the output arena is chosen, before the counted steps: not a kernel of the step)
    return ('rsa::' in name or 'rocprim' in name or 'rccl' in name.lower()) and 'placement_probe_kernel' not in name


def after_warmup(vals, steps, warm):
    calls = len(vals) / float(steps + warm)
    return (vals[int(round(calls * warm)):] if calls >= 0.99 else vals), calls


entry, lines = {}, [f'## {shape}']
c = db('trace')
steps, warm = steps_of('trace')
dom_name = None
if c is not None and steps:
    names = {}
    for name, st, en in c.execute('select name, start, end from kernels order by start'):
        if ours(name):
            names.setdefault(name, []).append(en - st)
    per_step = {}
    for name, durs in names.items():
        tail, calls = after_warmup(durs, steps, warm)
        per_step[name] = (sum(tail) / len(tail) / 1e3, calls, sum(tail) / 1e3 / steps)
    dom_name = max(per_step, key=lambda k: per_step[k][2])
    entry['kernel'] = dom_name.replace('void ', '')[:120]
    entry['avg_us'] = round(per_step[dom_name][0], 2)
    entry['launches'] = int(round(per_step[dom_name][1] * steps))
    entry['step_kernel_us'] = round(sum(v[2] for v in per_step.values()), 2)
    try:        # the per-allocation event times prof_shapes.py printed (output sets; the counted launches ran on the fastest)
        for line in open(os.path.join(out_dir, f'{shape}.trace.log')):
            if line.startswith('{"shape"') and 'out_set_us' in line:
                j = json.loads(line)
                entry['out_set_us'], entry['out_set_used'] = j['out_set_us'], j['out_set_used']
    except OSError:
        pass
    entry['dominant_share_of_step'] = round(per_step[dom_name][2] / max(entry['step_kernel_us'], 1e-9), 4)
    lines.append(f'# rocprofv3 --kernel-trace --stats -- python tools/prof_shapes.py {shape} {steps}   (kernels of the step, per step)')
    lines.append(f'{"kernel":100s} {"calls/step":>10s} {"avg_us":>10s} {"us/step":>10s}')
    for name, (avg, calls, tot) in sorted(per_step.items(), key=lambda kv: -kv[1][2]):
        lines.append(f'{name.replace("void ", "")[:100]:100s} {calls:10.2f} {avg:10.2f} {tot:10.2f}')
    lines.append(f'# step = {entry["step_kernel_us"]:.1f} us of kernel time; dominant kernel '
                 f'{entry["dominant_share_of_step"] * 100:.1f} % of it')
else:
    lines.append('no trace')

tot_bytes = {}
for sub, counter, mult in (('fetch', 'FETCH_SIZE', 2.0), ('write', 'WRITE_SIZE', 1.0)):
    c = db(sub)
    steps_p, warm_p = steps_of(sub)
    if c is None or not steps_p:
        lines.append(f'no {sub} pass')
        continue
    per_kernel = {}
    for name, value in c.execute('select kernel_name, value from counters_collection where counter_name=? order by id', (counter,)):
        if ours(name):
            per_kernel.setdefault(name, []).append(value)
    lines.append(f'# rocprofv3 --kernel-trace --pmc {counter}   (KB per launch, raw; bytes = KB x 1024 x {mult:g})')
    step_bytes = 0.0
    for name, vals in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1])):
        tail, calls = after_warmup(vals, steps_p, warm_p)
        avg = sum(tail) / len(tail)
        step_bytes += sum(tail) / steps_p * 1024 * mult
        lines.append(f'{name.replace("void ", "")[:100]:100s} launches={len(vals):4d} avg_KB={avg:14.1f} min_KB={min(tail):14.1f} max_KB={max(tail):14.1f}')
        if name == dom_name:
            entry[counter + '_KB_per_launch'] = round(avg, 1)
            tot_bytes[counter] = avg * 1024 * mult
    entry[counter + '_bytes_per_step_corrected'] = round(step_bytes)
if 'FETCH_SIZE' in tot_bytes and 'WRITE_SIZE' in tot_bytes:
    entry['hbm_bytes_per_launch'] = round(tot_bytes['FETCH_SIZE'] + tot_bytes['WRITE_SIZE'])
    lines.append(f'# dominant kernel HBM traffic per launch: 2 x FETCH_SIZE + WRITE_SIZE = {entry["hbm_bytes_per_launch"] / 1e9:.3f} GB')
# a partial re-collection starts from the committed summaries: entries / blocks of other shapes are kept
committed = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
ROUND = os.environ.get('ROUND', 'r04')
for name in (f'{ROUND}_kernel_profiles.json', f'{ROUND}_kernel_profiles.txt'):
    if not os.path.exists(os.path.join(dst, name)) and os.path.exists(os.path.join(committed, name)):
        shutil.copy(os.path.join(committed, name), os.path.join(dst, name))
path = os.path.join(dst, f'{ROUND}_kernel_profiles.json')
allj = json.load(open(path)) if os.path.exists(path) else {}
allj[shape] = entry
json.dump(allj, open(path, 'w'), indent=1, sort_keys=True)
tpath = os.path.join(dst, f'{ROUND}_kernel_profiles.txt')
blocks, order = {}, []
if os.path.exists(tpath):
    cur = None
    for line in open(tpath).read().split('\n'):
        if line.startswith('## '):
            cur = line[3:].strip()
            order.append(cur)
            blocks[cur] = []
        if cur is not None:
            blocks[cur].append(line)
if shape not in blocks:
    order.append(shape)
blocks[shape] = lines
open(tpath, 'w').write('\n\n'.join('\n'.join(blocks[k]).rstrip('\n') for k in order) + '\n')
print('\n'.join(lines))
print(json.dumps(entry))
shape_dir = os.path.join(out_dir, shape)
if os.path.isdir(shape_dir) and os.path.basename(shape_dir) == shape and shape:
    shutil.rmtree(shape_dir)
