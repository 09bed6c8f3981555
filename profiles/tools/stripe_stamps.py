"""Reads a FELICS_STRIPE_STAMPS dump (felics_api.cpp: run_stripe) and prints where a tile's time goes.

Stamps per ticket (100 MHz wall clock, thread 0): 0 ticket taken, 1 events partitioned (A done), 2 predecessor's token seen
(+ barrier), 3 own token stored (estimator rows written), 4 k of every event known, 5 tile packed and stored."""
import struct
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
nplanes, ntiles, per, _ = struct.unpack("<4I", raw[:16])
st = np.frombuffer(raw[16:], dtype=np.uint64).reshape(ntiles, nplanes, per).astype(np.int64)  # ticket = tile * nplanes + plane
good = (st[..., :6] > 0).all(axis=-1)  # a stamp that never landed (its tile had nothing to pack) reads zero
t0 = st[..., 0][good].min()
us = np.where(st > 0, (st - t0) / 100.0, np.nan)
names = ["A: load+classify+partition", "B1+wait for token", "B3: rows+walk+token", "B4: k per event", "C: pack+store"]
for i, n in enumerate(names):
    d = us[..., i + 1] - us[..., i]
    print("%-28s mean %7.2f us   p50 %7.2f   p95 %7.2f" % (n, np.nanmean(d), np.nanmedian(d), np.nanpercentile(d, 95)))
tot = us[..., 5] - us[..., 0]
print("tile total                   mean %7.2f us" % np.nanmean(tot))
print("kernel span %.1f us; tiles %d x planes %d" % (np.nanmax(us[..., 5]), ntiles, nplanes))
# hand-off: token stored by tile t (stamp 3) -> token seen by tile t + 1 (stamp 2)
if ntiles > 1:
    hop = us[1:, :, 2] - us[:-1, :, 3]
    ready = us[1:, :, 1] - us[:-1, :, 3]  # > 0: successor was ready before the token came (it waited)
    print("hop (token stored -> seen by the next tile, incl. barrier): mean %.2f us p50 %.2f p95 %.2f" % (np.nanmean(hop), np.nanmedian(hop), np.nanpercentile(hop, 95)))
    print("successor finished A before the token was stored in %.0f %% of hand-offs" % (100.0 * np.nanmean(ready < 0)))
    chain = us[1:, :, 3] - us[:-1, :, 3]
    print("token-to-token period per plane: mean %.2f us  (x %d tiles = %.0f us)" % (np.nanmedian(chain), ntiles, np.nanmedian(chain) * ntiles))

# inside B3, per wave (stamps 16 + wave * 4 + k): 0 rows loaded, 1 one-block contexts replayed, 2 walks done, 3 stores drained
if per >= 80:
    wv = us[..., 16:80].reshape(ntiles, nplanes, 16, 4)
    begin = us[..., 2][..., None]
    for k, n in enumerate(["rows loaded", "one-block contexts replayed", "walks done", "row stores drained"]):
        d = wv[..., k] - begin
        print("B3 wave stamp %-28s after token: slowest wave median %6.2f us, mean over waves %6.2f" % (n, np.nanmedian(np.nanmax(d, axis=-1)), np.nanmean(d)))
    walk = wv[..., 2] - wv[..., 1]
    print("walk time per wave (median over tiles):", np.round(np.nanmedian(walk.reshape(-1, 16), axis=0), 2))

if per >= 80:
    def seg(name, a, b):
        d = us[..., b] - us[..., a]
        print("  %-40s p50 %6.2f us" % (name, np.nanmedian(d)))
    print("finer split (thread 0):")
    seg("A1 load span", 0, 6)
    seg("A2 classify + count", 6, 7)
    seg("A3 scans", 7, 8)
    seg("A4 ring + rank + scatter", 8, 1)
    seg("C1 bit strings", 4, 9)
    seg("C2 look-back", 9, 10)
    seg("C3 window + store", 10, 5)
