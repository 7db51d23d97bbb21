"""Cell graph, enumeration A1: of the (own cell, neighbour row) items of the forward half of the 5×5×5 neighbourhood, how many find no cell at all in their row's window x − 2 … x + 2
(min / median / max over the streams of a workload, from the cell keys the device built)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
ROWS = [(0, 0), (0, 1), (1, -1), (1, 0), (1, 1), (0, 2), (1, -2), (1, 2), (2, -2), (2, -1), (2, 0), (2, 1), (2, 2)]
for wl in sys.argv[1:] or ["hdl64_b64", "hdl64_urban_b64"]:
    p = kitti_params(1)
    leg = bench.Leg(engine, synth, shard, p, wl, 0, 0, 3, 16)
    for _ in range(2):
        leg.step()
    cfg = leg.batch.debug_config(); nx, nz = cfg["nx"], cfg["nz"]
    empty, near_empty, cells = [], [], []
    for s in range(leg.B):
        n = leg.batch.stage_counts(s)["n_occ"]
        key = leg.batch.debug_read("ckey", s, np.int32, n).astype(np.int64)
        occ = set(key.tolist())
        x, row = key % nx, key // nx
        z, y = row % nz, row // nz
        tot = e = ne = 0
        for ri, (dy, dz) in enumerate(ROWS):
            zz = z + dz
            ok = (zz >= 0) & (zz < nz)
            base = ((y + dy) * nz + zz) * nx + x
            hit = np.zeros(n, bool)
            for dx in range(-2, 3):
                if dy == 0 and dz == 0 and dx <= 0:
                    continue
                hit |= ok & np.isin(base + dx, key)
            tot += n; e += int((~hit).sum())
            if ri < 5:
                ne += int((~hit).sum())
        empty.append(e / tot); near_empty.append(ne / (5 * n)); cells.append(n)
    print(json.dumps({"workload": wl, "cells": [min(cells), int(np.median(cells)), max(cells)], "items_without_a_neighbour": [round(min(empty), 3), round(float(np.median(empty)), 3), round(max(empty), 3)],
                      "of_the_five_near_rows": [round(min(near_empty), 3), round(float(np.median(near_empty)), 3), round(max(near_empty), 3)]}))
    leg.close()
