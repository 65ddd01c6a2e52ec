#!/bin/bash
# AddressSanitizer + UBSan on the CPU-side code (GPU sanitizers are not available on the pool): the oracle
# (test infrastructure) and the product's host-side C++ (fcpt_host.cpp: split, grid, initial conditions) over the
# fuzzer's draws.  usage: bash profiles/tools/sanitize_cpu.sh   (here, no GPU needed)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=${TMPDIR:-/tmp}
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -std=c99 -fopenmp -shared \
    -I$R/include -o $T/liborc_asan.so $R/oracle/fargo_oracle.c -lm
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -std=c++17 -shared -I$R/include \
    -I$R/fargocpt_amd/csrc -o $T/libhost_asan.so $R/fargocpt_amd/csrc/fcpt_host.cpp
cat > $T/sanitize_run.py <<PY
import sys, ctypes, numpy as np
sys.path.insert(0, "$R")
from fargocpt_amd import binding as B, setups
from fargocpt_amd.binding import Library
from tests.util import run_pair
import tests.test_gpu_fuzz as T
H = Library(ctypes.CDLL("$T/libhost_asan.so"), "fcpt_")
O = Library(ctypes.CDLL("$T/liborc_asan.so"), "orc_")
host = orc = 0
for seed in list(range(0, 160)) + list(range(50000, 50300)):
    T.WIDE = seed >= 50000
    d, nslabs, planet = T.draw(H, seed)
    extra = dict(T._EXTRA)
    for rank in range(nslabs):
        dd = d.copy(); dd.rank, dd.nranks = rank, nslabs
        try:
            H.split_domain(dd)
        except B.FcptError:
            continue
        radii = H.radii(dd)
        if np.isfinite(radii).all():
            H.initial_fields(dd, radii); host += 1
    if seed % 5 == 0:
        try:
            run_pair(O, O, d, 4, bodies=setups.jupiter_bodies(d) if planet else None, nslabs=(nslabs, 0), **extra); orc += 1
        except B.FcptError:
            pass
print("clean: host-side calls", host, "oracle runs", orc)
PY
OMP_NUM_THREADS=2 ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python3 $T/sanitize_run.py
