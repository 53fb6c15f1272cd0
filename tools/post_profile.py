"""Where does a post launch's time go?  (LVBGPU_POST_PROFILE: clock stamps of its workgroups, by role)
One chain (or R) at 500 x 50k: steps of a few candidates with one accepted each time, the post launch of every step
inspected: per role - table rebuild, commit walk, generator - when its workgroups started and ended relative to the launch's
first stamp.
  gpurun -- python tools/post_profile.py [R] [B] [beside]
beside > 0: another context (a lane) has a scoring walk of that many candidates on the device while each post launch runs.
"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["LVBGPU_POST_PROFILE"] = "1"
import numpy as np
from lvb_amd import api, host
from tests.synth import treelike_rows

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
BESIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, m = 500, 50000
rows, min_len = host.prepare_alignment(treelike_rows(n, m, 3))
ctx = api.FitchContext(text_rows=rows)
ctx.set_chains(R)
for c in range(R):
    ctx.select_chain(c)
    host.HostTree(n, seed=100 + c).upload(ctx)
other = None
if BESIDE:
    other = ctx.fork()
    host.HostTree(n, seed=999).upload(other)
buf = np.zeros(4065, dtype=np.uint64)
phases = []
names = {1: "rebuild", 2: "commit", 3: "generator"}
acc = {}
slot = 0
draws = [(c, B, 1, 1000 + c) for c in range(R)]
ctx.chains_submit(slot, draws)
for step in range(60):
    lens = ctx.chains_collect(slot, [B] * R)
    ctx.chains_commit([(c, int(np.argmin(lens[c]))) for c in range(R)])
    slot ^= 1
    if other is not None:
        cnt = other.chains_submit(0, [(0, BESIDE, 1, 77 + step)])      # the other lane's walk: on the device when the post launch arrives
    ctx.chains_submit(slot, [(c, B, 1, 5000 + 10 * step + c) for c in range(R)])      # post launch: commits + rebuilds + this generator
    if other is not None:
        ctx.synchronize()
        other.chains_collect(0, cnt)
    if step >= 20:
        ctx._chk(ctx.lib.lvbgpu_debug_post_stamps(ctx.h, buf.ctypes.data))
        nb = int(buf[0])
        rec = buf[1:1 + 4 * min(nb, 1000)].reshape(-1, 4)
        rec = rec[rec[:, 0] > 0]
        t0 = rec[:, 1].min()
        ph = buf[4001:4001 + 8].astype(np.int64)
        if ph[0] > 0:
            phases.append([(int(ph[k]) - int(ph[0])) / 100.0 if ph[k] > 0 else np.nan for k in range(6)])
        for role in np.unique(rec[:, 0]):
            r = rec[rec[:, 0] == role]
            acc.setdefault(int(role), []).append((len(r), (r[:, 1].min() - t0) / 100.0, (r[:, 3].max() - t0) / 100.0,
                                                  float(np.mean(r[:, 3] - r[:, 1])) / 100.0))
print(f"R={R} B={B} beside a walk of {BESIDE}: post launches inspected: {len(next(iter(acc.values())))}")
for role, v in sorted(acc.items()):
    a = np.array(v)
    print(f"  {names.get(role, role):10s} workgroups {a[:,0].mean():6.1f}  first start {a[:,1].mean():6.2f} us  last end {a[:,2].mean():6.2f} us  "
          f"mean workgroup time {a[:,3].mean():6.2f} us")
if phases:
    a = np.nanmean(np.array(phases), axis=0)
    print('  rebuild phases of workgroup 0 (us from its start): tables loaded %.2f, parents + leaves below %.2f, root-ward walks %.2f, drawn %.2f, copied out %.2f' % tuple(a[1:6]))
ctx.close()
