#!/usr/bin/env python3
"""Start / end times of the wavefronts of k_source_march / k_source_march_adi (a library built with -DSM_TRACE leaves
them, in 10 ns ticks of s_memrealtime, in the temperature grid, which neither kernel touches): which wavefronts end
when, and how many are resident over the kernel's duration.

    make -C fargocpt_amd/csrc alt ALTNAME=smtrace EXTRA=-DSM_TRACE
    FCPT_LIB_PATH=$PWD/fargocpt_amd/libfargocpt_hip_smtrace.so python profiles/tools/wave_trace_source.py [isothermal|ideal] [nr nphi]
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import fargocpt_amd
from fargocpt_amd import binding as B, driver, setups
lib = fargocpt_amd.load()
adi = len(sys.argv) > 1 and sys.argv[1] == "ideal"
NR, NPHI = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 4096)
d = setups.planet_disk(lib, NR, NPHI, adiabatic=adi)
ctx = driver.make_context(lib, d, bodies=setups.jupiter_bodies(d))
for _ in range(2):
    ctx.calculate_timestep(ctx.cfl())
ctx.run_steps(200)
ctx.synchronize()
# raw copy of the grid (a download would materialise the lazily derived temperature of the ideal EOS over the records)
ptr, count = ctx.device_ptr(B.F_TEMPERATURE)
t = np.zeros(count)
hip = C.CDLL("libamdhip64.so")
assert hip.hipMemcpy(t.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(t.nbytes), C.c_int(2)) == 0
segs = (NPHI + 58) // 59
rec = t[:4 * (t.size // 4)].reshape(-1, 4)
good = (rec[:, 3] >= 1) & (rec[:, 3] <= NR + 1) & (rec[:, 2] >= 0) & (rec[:, 2] < rec[:, 3]) & (rec[:, 1] > rec[:, 0]) & (rec[:, 0] > 0) & (rec[:, 3] == np.floor(rec[:, 3]))
n = int(np.argmin(good)) if not good.all() else good.size
rec = rec[:n]
st, en, k0, k1 = rec[:, 0], rec[:, 1], rec[:, 2].astype(int), rec[:, 3].astype(int)
print("waves", n, "segments", segs, "chunks", n // segs, "rings per chunk", k1[0] - k0[0])
t0 = st.min()
s0, e0 = (st - t0) / 100.0, (en - t0) / 100.0
dur = e0 - s0
print("kernel span us %.1f" % e0.max())
print("duration us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
print("start us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(s0, q) for q in (50, 90, 99, 100)))
print("end us: p1 %.1f p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(e0, q) for q in (1, 10, 25, 50, 75, 90, 99, 100)))
w = np.arange(n)
# by position in the dispatch order (xcd_block deals contiguous ranges of workgroups to the XCDs: range = XCD)
per = (n + 7) // 8
print("per XCD range, by eighth of its dispatch order: mean end us")
for x in range(8):
    m = (w // per) == x
    pos = (w[m] - x * per) * 8 // per
    print("  xcd %d:" % x, " ".join("%6.1f" % e0[m][pos == q].mean() for q in range(8) if (pos == q).any()), "  start of the last eighth %.1f" % s0[m][pos == pos.max()].mean())
if ctx.get_option("source_graded") != 0 and len(ctx.source_chunks()):
    # table order: workgroup b = blockIdx.x runs on XCD b % 8
    xcd = (w // 4) % 8
    print("rank-matched table: per XCD wavefronts, rings marched, mean / max end us")
    for x in range(8):
        m = xcd == x
        print("  xcd %d: %5d wavefronts %7d rings  end mean %.1f max %.1f" % (x, m.sum(), (k1[m] - k0[m]).sum(), e0[m].mean(), e0[m].max()))
ch = w // segs
print("per chunk: mean end us")
print(" ".join("%.0f" % e0[ch == c].mean() for c in range(n // segs)))
ev = np.concatenate([np.stack([s0, np.ones(n)], 1), np.stack([e0, -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0])]
occ = np.cumsum(ev[:, 1])
for tq in range(0, min(int(e0.max()) + 1, 1000), 10):
    k = np.searchsorted(ev[:, 0], tq)
    print("  t %4d us resident %5d" % (tq, occ[k - 1] if k > 0 else 0))
ctx.close()
