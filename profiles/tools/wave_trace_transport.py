#!/usr/bin/env python3
"""Start / end times of the wavefronts of k_transport_fused (a library built with -DTF_TRACE leaves them, in 10 ns
ticks of s_memrealtime, in the temperature grid): which chunks are slow, and where the idle wavefront slots come from.

    make -C fargocpt_amd/csrc alt ALTNAME=tftrace EXTRA=-DTF_TRACE
    FCPT_LIB_PATH=$PWD/fargocpt_amd/libfargocpt_hip_tftrace.so python profiles/tools/wave_trace_transport.py [isothermal|ideal] [nr nphi]
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
tiles = (NPHI + 52) // 53
rec = t[:4 * (t.size // 4)].reshape(-1, 4)
good = (rec[:, 3] >= 1) & (rec[:, 3] <= NR) & (rec[:, 2] >= 0) & (rec[:, 2] < rec[:, 3]) & (rec[:, 1] > rec[:, 0]) & (rec[:, 0] > 0) & (rec[:, 3] == np.floor(rec[:, 3]))
n = int(np.argmin(good)) if not good.all() else good.size  # the records are contiguous from the start of the grid
n -= n % tiles
rec = rec[:n]
st, en, q0, q1 = rec[:, 0], rec[:, 1], rec[:, 2].astype(int), rec[:, 3].astype(int)
ok = (en > 0) & (st > 0)
print("waves", n, "valid", ok.sum(), "chunks", n // tiles)
t0 = st[ok].min()
s0, e0 = (st - t0) / 100.0, (en - t0) / 100.0
dur = e0 - s0
print("kernel span us %.1f" % e0[ok].max())
print("duration us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f; sum/4096 = %.1f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum() / 4096))
print("start us: p25 %.1f p50 %.1f p75 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(s0, q) for q in (25, 50, 75, 90, 99, 100)))
print("end us: p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(e0, q) for q in (1, 10, 50, 90, 99, 100)))
ch = np.arange(n) // tiles
print("per chunk (dispatch order): rings, start / duration / end")
for c in range(n // tiles):
    m = ch == c
    print("  chunk %3d xcd %d rings %4d..%4d (%2d)  start %6.1f  dur %6.1f  end %6.1f" % (c, c % 8, q0[m][0], q1[m][0] - 1, q1[m][0] - q0[m][0], s0[m].mean(), dur[m].mean(), e0[m].mean()))
# resident wavefronts over time
ev = np.concatenate([np.stack([s0, np.ones(n)], 1), np.stack([e0, -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0])]
occ = np.cumsum(ev[:, 1])
for tq in range(0, min(int(e0.max()) + 1, 1000), 10):
    k = np.searchsorted(ev[:, 0], tq)
    print("  t %4d us resident %5d" % (tq, occ[k - 1] if k > 0 else 0))
ctx.close()
