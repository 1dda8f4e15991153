#!/usr/bin/env python3
"""What does a hipExtStreamCreateWithCUMask mask select on an MI355X (SPX mode, 8 XCDs x 32 CUs)?  For a set of masks: which XCC ids and how many
distinct CUs a resident 2048-block grid lands on (HW_REG_XCC_ID / HW_REG_HW_ID per block, ns2vc_debug_placement).  r5."""
import ctypes as C, os, sys, functools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ns2vc_amd import _lib
from ns2vc_amd._lib import check
from ns2vc_amd.engine import Stream
print = functools.partial(print, flush=True)
lib = _lib.load()


def place(st, n=2048, spin=300):
    out = (C.c_uint32 * (2 * n))()
    check(lib.ns2vc_debug_placement(st.ptr if st else None, n, spin, out), "placement")
    a = np.frombuffer(out, dtype=np.uint32).reshape(n, 2)
    return a[:, 0].copy(), a[:, 1].copy()


masks = {"none (plain stream)": None, "bits 0-31": range(32), "bits 32-63": range(32, 64), "bits 0-127": range(128), "bits 128-255": range(128, 256),
         "every 8th bit (0, 8, ..)": range(0, 256, 8), "bits = 1 mod 8": range(1, 256, 8), "all but every 8th": [b for b in range(256) if b % 8],
         "bit 0": [0], "bit 1": [1], "bit 8": [8], "bit 255": [255], "bits 0-223": range(224), "bits 224-255": range(224, 256),
         "4 per XCD if interleaved (b // 8 >= 28)": [b for b in range(256) if b // 8 >= 28], "4 per XCD if XCD-major (b % 32 >= 28)": [b for b in range(256) if b % 32 >= 28]}
for name, m in masks.items():
    st = Stream(cu_mask=list(m)) if m is not None else Stream()
    x, h = place(st)
    hist = np.bincount(x, minlength=8).tolist()
    cus = len(set(zip(x.tolist(), ((h >> 8) & 0x7f).tolist())))      # distinct (XCC, HW_ID.{se_id, sh_id, cu_id})
    first = x[:16].tolist()
    print(f"{name:42s} blocks per XCC {hist}  distinct (XCC, CU) pairs {cus:3d}  first 16 blocks -> XCC {first}")
