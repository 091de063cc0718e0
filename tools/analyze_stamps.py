#!/usr/bin/env python3
"""Reads the stamp file written under TTSDEC_STAMPS (csrc/api.hip) and prints, per two-role launch of the last step:
role placement over CUs and the start / gate / end times (us, relative to the launch's first start).
A second argument "merged": the one-launch step (option overlap = 3) - both role pairs are timed from the launch's first stamp."""
import sys
from collections import Counter

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 1024, 8)
merged = len(sys.argv) > 2 and sys.argv[2] == "merged"
t0_kind0 = None
for kind, name in ((0, "frame || lstm_att"), (1, "attention || lstm_dec")):
    s = a[kind]
    used = s[:, 2] > 0
    if not used.any():
        continue
    s = s[used]
    role = (s[:, 0] >> 32).astype(int)
    hw = (s[:, 0] & 0xFFFFFFFF).astype(int)
    cu = ((s[:, 1].astype(int) & 0xF) << 16) | (hw & 0xFF00)  # xcc | se/sh/cu bits of HW_ID
    t0 = int(s[:, 2].min())
    if kind == 0:
        if s[:, 7].any():
            t0 = min(t0, int(s[:, 7][s[:, 7] > 0].min()))
        if a.shape[0] > 2 and (a[2][:, 2] > 0).any():
            t0 = min(t0, int(a[2][:, 2][a[2][:, 2] > 0].min()))  # the projection head role starts the launch
        t0_kind0 = t0
    elif merged and t0_kind0 is not None:
        t0 = t0_kind0
    us = lambda col: (s[:, col].astype(np.int64) - t0) / 100.0
    print(f"== {name}: {used.sum()} workgroups stamped")
    for r, rn in ((0, "producer"), (1, "lstm")):
        m = role == r
        if not m.any():
            continue
        st, en = us(2)[m], us(5)[m]
        print(f"  {rn:8s} n={m.sum():4d} start {st.min():6.2f}..{st.max():6.2f}  end {en.min():6.2f}..{en.max():6.2f} (mean {en.mean():6.2f})")
        if r == 0 and kind == 0:  # frame role: 3 = previous step's frame finished, 4 = layer 0 done, 6 = x_pre stores issued
            for col, what in ((7, "entry (before the wait)"), (3, "frame(t-1) finished"), (4, "layer 0 done"), (6, "x_pre stores issued")):
                v = us(col)[m]
                ok = s[:, col][m] > 0
                if ok.any():
                    print(f"           {what:20s} {v[ok].min():6.2f}..{v[ok].max():6.2f} (mean {v[ok].mean():6.2f})")
        if r == 0 and kind == 1 and merged:
            v = us(6)[m]
            ok = s[:, 6][m] > 0
            if ok.any():
                print(f"           role entry (before the query tiles) {v[ok].min():6.2f}..{v[ok].max():6.2f} (mean {v[ok].mean():6.2f})")
        if r == 1:
            gi, go = us(3)[m], us(4)[m]
            ok = s[:, 3][m] > 0
            print(f"           gate reached {gi[ok].min():6.2f}..{gi[ok].max():6.2f} (mean {gi[ok].mean():6.2f})  passed {go[ok].min():6.2f}..{go[ok].max():6.2f} (mean {go[ok].mean():6.2f})")
            if not merged and (s[:, 6][m] > 0).any() and (s[:, 7][m] > 0).any():  # K-loop progress marks
                t16, t32 = us(6)[m], us(7)[m]
                st_ = us(2)[m]
                print(f"           K tile 16 at {t16.min():6.2f}..{t16.max():6.2f} (mean {t16.mean():6.2f}), tile 32 at {t32.min():6.2f}..{t32.max():6.2f} (mean {t32.mean():6.2f})")
                if a.shape[0] > kind + 3:  # shader-clock readings at the same two points
                    c = a[kind + 3][used][m]
                    okc = (c[:, 0] > 0) & (c[:, 1] > c[:, 0])
                    if okc.any():
                        ghz = (c[okc, 1] - c[okc, 0]).astype(np.float64) / ((t32[okc] - t16[okc]) * 100.0) * 0.1
                        print(f"           in-kernel clock over K tiles 16..32: median {np.median(ghz):5.3f} GHz (min {ghz.min():5.3f}, max {ghz.max():5.3f}); "
                              f"{(t32[okc] - t16[okc]).mean() / 16 * 1e3:6.1f} ns = {np.median((c[okc, 1] - c[okc, 0]) / 16.0):7.1f} cycles per K tile")
                slow = np.argsort(-gi)[:32]
                fast = np.argsort(gi)[:32]
                for nm, idx in (("32 slowest", slow), ("32 fastest", fast)):
                    print(f"             {nm}: start {st_[idx].mean():5.2f}  ->t16 {(t16[idx]-st_[idx]).mean():5.2f}  ->t32 {(t32[idx]-t16[idx]).mean():5.2f}  ->gate {(gi[idx]-t32[idx]).mean():5.2f} us")
            ok2 = s[:, 6][m] > 0
            if merged and kind == 1 and ok2.any():  # (one-launch step: the decoder LSTM's h_att segment)
                g2i, g2o = us(6)[m], us(7)[m]
                print(f"           h_att gate reached {g2i[ok2].min():6.2f}..{g2i[ok2].max():6.2f} (mean {g2i[ok2].mean():6.2f})  passed {g2o[ok2].min():6.2f}..{g2o[ok2].max():6.2f} (mean {g2o[ok2].mean():6.2f})")
    per_cu = Counter()
    for c, r in zip(cu, role):
        per_cu[(c, r)] += 1
    lstm_per_cu = Counter(v for (c, r), v in per_cu.items() if r == 1)
    prod_per_cu = Counter(v for (c, r), v in per_cu.items() if r == 0)
    both = sum(1 for c in set(cu) if per_cu.get((c, 0), 0) and per_cu.get((c, 1), 0))
    print(f"  CUs seen {len(set(cu))}; lstm workgroups per CU {dict(lstm_per_cu)}; producer workgroups per CU {dict(prod_per_cu)}; CUs holding both roles {both}")

if a.shape[0] > 2 and (a[2][:, 2] > 0).any() and t0_kind0 is not None:
    s = a[2][a[2][:, 2] > 0]
    us = lambda col: (s[:, col].astype(np.int64) - t0_kind0) / 100.0
    print(f"== projection head role of the frame || lstm_att launch: {len(s)} workgroups (us from the launch's first stamp)")
    for col, what in ((2, "entry"), (3, "control block + operands arrived"), (4, "partial tile reduced"), (5, "signalled")):
        v = us(col)
        print(f"           {what:34s} {v.min():6.2f}..{v.max():6.2f} (mean {v.mean():6.2f})")
