#!/usr/bin/env python3
"""Start / end times of the wavefronts of k_source_march (a library built with -DSM_TRACE leaves them, in 10 ns
ticks of s_memrealtime, in the temperature grid): where the idle wavefront slots of the kernel come from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import fargocpt_amd
from fargocpt_amd import binding as B, driver, setups
lib = fargocpt_amd.load()
d = setups.planet_disk(lib, 2048, 4096)
ctx = driver.make_context(lib, d, bodies=setups.jupiter_bodies(d))
for _ in range(2):
    ctx.calculate_timestep(ctx.cfl())
ctx.run_steps(200)
ctx.synchronize()
t = ctx.download(B.F_TEMPERATURE).ravel()
segs = (4096 + 58) // 59
rows = 24
chunks = (2049 + rows - 1) // rows
n = segs * chunks
st, en = t[0:2 * n:2], t[1:2 * n:2]
st = np.full_like(en, en[en > 0].min() - 7000.0)  # (end times only: the start is not recorded)
ok = en > 0
st, en = st[ok], en[ok]
t0 = st.min()
print("waves", n, "valid", ok.sum(), "kernel span us", (en.max() - t0) / 100.0)
dur = (en - st) / 100.0
print("duration us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
s0 = (st - t0) / 100.0
print("start us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % (np.percentile(s0, 50), np.percentile(s0, 90), np.percentile(s0, 99), s0.max()))
e0 = (en - t0) / 100.0
print("end us: p1 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (np.percentile(e0, 1), np.percentile(e0, 10), np.percentile(e0, 50), np.percentile(e0, 90), e0.max()))
# by chunk
w = np.arange(n)[ok]
ch = w // segs
for c in (0, 1, chunks // 2, chunks - 2, chunks - 1):
    m = ch == c
    if m.any():
        print("chunk", c, "start %.1f end %.1f dur %.1f" % (s0[m].mean(), e0[m].mean(), dur[m].mean()))
print("per chunk mean end:", " ".join("%.0f" % e0[ch == c].mean() for c in range(chunks)))
sg = w % segs
print("per seg mean end:", " ".join("%.0f" % e0[sg == q].mean() for q in range(segs)))
blk = w // 4
print("per block-in-256 (block index mod 256) mean end, first 32:", " ".join("%.0f" % e0[(blk % 256) == q].mean() for q in range(32)))
# by XCD-range (contiguous blocks)
for x in range(8):
    m = (w * 8 // n) == x
    print("range", x, "start %.1f end %.1f dur %.1f" % (s0[m].mean(), e0[m].mean(), dur[m].mean()))
ctx.close()
