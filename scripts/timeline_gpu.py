"""In-kernel timeline of loam_iterate_kernel (pcr_params.record_timeline = 1): where the microseconds of one launch go."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth, pcr
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
p = pcr.default_params(loam_iters=10, loam_early_exit=0)
p.record_timeline = 1
reg = LoamRegister(params=p)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
for i in range(3):
    pose = T0.copy(); reg.scan2Map(ds, dm, pose)
tl = reg.timeline()
names = ['entry', 'prologue', 'posted', 'searched', 'plane', 'accum', 'stored']
np.set_printoptions(precision=2, suppress=True, linewidth=200)
print('per launch: mean over blocks of each stamp (us since the launch\'s first block entry); last column = max stored')
for k in range(tl.shape[0]):
    t = tl[k]
    ok = t[:, 6] > 0
    print(k, ' '.join(f'{n}={t[ok, i].mean():6.2f}' for i, n in enumerate(names)), f'max_stored={t[ok, 6].max():6.2f}', f'entry_spread={t[ok, 0].max():5.2f}')
print('stage durations (mean over blocks):')
for k in range(tl.shape[0]):
    t = tl[k]; ok = t[:, 6] > 0
    d = np.diff(t[ok][:, :7], axis=1).mean(axis=0)
    print(k, ' '.join(f'{names[i + 1]}:{d[i]:6.2f}' for i in range(6)), f'fold_done_at={t[ok, 7].mean() - t[ok, 0].mean():5.2f} after entry')
for L in (4, 5, 6):
    print(f'slowest blocks of launch {L} (stage durations):')
    t = tl[L]; d = np.diff(t[:, :7], axis=1)
    for b in np.argsort(-t[:, 6])[:4]:
        print('   ', b, f'entry={t[b,0]:5.2f} stored={t[b,6]:6.2f}', ' '.join(f'{names[i + 1]}:{d[b, i]:6.2f}' for i in range(6)))
t = tl[6]
print('launch 6 stored-time percentiles', np.percentile(t[:, 6], [0, 10, 50, 90, 99, 100]))
print('dense search of launch 0 (thread 0 of each block): posted -> ranges in LDS -> first chunk -> stream done -> searched')
t = tl[0]
ok = t[:, 10] > 0
for a, b, n in ((2, 8, 'ranges'), (8, 9, 'first chunk'), (9, 10, 'stream'), (10, 3, 'ties + exchange')):
    print(f'  {n:16s} {np.mean(t[ok, b] - t[ok, a]):6.2f} us')
print('prologue of launch 6: entry -> partial sums folded -> system solved -> pose updated (prologue end)')
t = tl[6]
print(f'  fold {np.mean(t[:, 7] - t[:, 0]):5.2f} us, solve {np.mean(t[:, 11] - t[:, 7]):5.2f} us, exp + pose update {np.mean(t[:, 1] - t[:, 11]):5.2f} us')
