"""Where a K step of k_dots_filter spends its cycles (ablation build, ISC_DEBUG_MODE=22: s_memtime stamps in the loop).

    ISC_LIB=imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1 ISC_DEBUG_MODE=22 python scripts/stamp_search.py [N] [Q]

Every wave adds up, per K step, [reads + MFMA issue] [counted vmcnt wait] [barrier] and leaves the sums at the start of the
(unused in this mode) survivor segments of the workspace.  The stamps cost cycles and forbid overlaps: read the SHARES."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagescry_amd import EmbeddingBank

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
bank = torch.nn.functional.normalize(torch.randn(n, 768, generator=g, device=dev), dim=1).half()
queries = torch.randn(q, 768, generator=g, device=dev).half()
eb = EmbeddingBank(bank, dtype=torch.float16, normalize=False)
for _ in range(3):
    eb.search(queries, 10)
torch.cuda.synchronize()
ws = next(iter(eb._workspaces.values()))
qpad = -(-q // 64) * 64 if q <= 128 else -(-q // 256) * 256
kp = 16
al = lambda v: -(-v // 256) * 256
off = al(qpad * 4) + 2 * al(qpad * kp * 4) + 2 * al(qpad * 4)  # tau, carry_s, carry_r, carry_n, qflag -> seg_ent
qtiles = qpad // (64 if q <= 128 else 256)
wgs = 256 // qtiles * qtiles
raw = ws[off : off + wgs * 8 * 4 * 8].view(torch.int64).view(-1, 8, 4).cpu().numpy().astype(np.float64)
raw = raw[raw[:, 0, 3] > 0]
steps = raw[:, :, 3]
per = raw[:, :, :3] / steps[:, :, None]
tot = per.sum(axis=2)
print(f"N={n} Q={q}: {raw.shape[0]} workgroups, {steps.mean():.0f} steps per wave in the last launch")
print(f"cycles per K step and wave (mean over workgroups): total {tot.mean():.0f}")
for w in range(8):
    c, v, b = per[:, w, 0].mean(), per[:, w, 1].mean(), per[:, w, 2].mean()
    t = c + v + b
    print(f"  wave {w}: reads+MFMA issue {c:7.0f} ({100*c/t:4.1f} %)  vmcnt wait {v:6.0f} ({100*v/t:4.1f} %)  barrier {b:6.0f} ({100*b/t:4.1f} %)")
