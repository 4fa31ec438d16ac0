"""profiles/r03_kernel_profiles.txt -> the per-topic extracts VERDICT r2 asked for by name (same blocks, nothing new):
r03_shard_kernels.txt (sharded step + sharded training step), r03_small_batch.txt (B = 4096 / 16 384),
r03_train_kernels.txt (in-place SGD step, forward + backward step).
usage: python tools/split_profiles.py [profiles_dir]"""
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
blocks, cur = {}, None
for line in open(os.path.join(d, 'r03_kernel_profiles.txt')).read().split('\n'):
    if line.startswith('## '):
        cur = line[3:].strip()
        blocks[cur] = []
    if cur is not None:
        blocks[cur].append(line)
TOPICS = {
    'r03_shard_kernels.txt': ('sharded_world1_step', 'sharded_world1_train'),
    'r03_small_batch.txt': ('N1e7_popular_n64_B4096', 'N1e7_popular_n64_B16384'),
    'r03_train_kernels.txt': ('sgd_step_N1e7_popular_n64_B65536', 'train_step_N1e7_popular_n64_B65536'),
}
for name, shapes in TOPICS.items():
    head = f'# extract of r03_kernel_profiles.txt (tools/split_profiles.py): {", ".join(shapes)}\n\n'
    body = '\n\n'.join('\n'.join(blocks[s]).rstrip('\n') for s in shapes if s in blocks)
    open(os.path.join(d, name), 'w').write(head + body + '\n')
    print(name, [s for s in shapes if s in blocks])
