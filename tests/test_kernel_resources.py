"""Registers, scratch and LDS of the hot kernels as hipcc reports them for gfx950 (-Rpass-analysis=kernel-resource-usage; runs without a GPU).
The frame pipeline's throughput hangs on how many workgroups of each kernel a CU holds, and the compiler's register count moves with unrelated edits:
round 4 lost 7 % when a loop took k_cg_slab from 105 to 157 VGPRs (one workgroup per CU instead of two) and 5 % when k_split went from 126 to 150 —
both silently.  This test pins what the design relies on."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dynamicslamtool_amd", "csrc", "mor_kernels.hip")

# kernel: (max VGPRs, max scratch bytes per lane, max LDS bytes per workgroup)
LIMITS = {
    "k_split<0>": (80, 0, 28 * 1024),           # three 512-thread workgroups per CU (round 6: eight waves × four rows — 2048-record tiles at 73 VGPRs; 48 bytes of LDS per thread are the compiler's: the tiles' class arrays promoted from private memory)
    "k_split<1>": (80, 0, 28 * 1024),           # pass A of the voxel ground variant: the same tiles and the same three workgroups per CU
    "k_rhist": (64, 0, 2 * 1024),                   # eight waves per SIMD for the histogram loop (round 6: the offsets scan with a 64-tile column in registers took 98 VGPRs and 1 % of the variant)
    "k_gridcount": (64, 0, 36 * 1024),
    "k_gridplace": (80, 0, 24 * 1024),
    "k_cellboxes": (128, 0, 1024),
    "k_cg_slab": (128, 0, 78 * 1024),               # two 512-thread workgroups per CU (registers and LDS)
    "k_clusters": (128, 0, 20 * 1024),
    "k_score_fast": (64, 0, 26 * 1024),             # two 1024-thread workgroups per CU
    "k_score_nb": (128, 0, 26 * 1024),              # two 512-thread workgroups per CU
    "k_score_pde": (64, 0, 1024),
    "k_track_filter": (128, 0, 12 * 1024),
    "k_out": (64, 0, 4 * 1024),
    "k_g2_cov": (96, 16, 1024),                     # voxel ground variant: FIVE waves per SIMD asked of the compiler (three spilled words; at four waves the pipeline lost 2 %), no LDS
}


def _usage(tmp_path):
    from dynamicslamtool_amd import build
    out = subprocess.run([build._hipcc()] + [f for f in build.HIP_FLAGS if f not in ("-shared",)] + ["-c", "-Rpass-analysis=kernel-resource-usage", "-x", "hip", SRC, "-o", str(tmp_path / "k.o")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(MorDev.*$", "", re.sub(r"^void ", "", name))
            if not cur.startswith("k_split"):
                cur = re.sub(r"<\d+>", "", cur)      # k_cg_slab<1024> → k_cg_slab (k_split<0|1|2>: the crop variant's split, pass A and pass B of the voxel ground variant)
            res[cur] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                res[cur][key] = int(m.group(1))
    return res


def test_hot_kernels_keep_their_occupancy(tmp_path):
    res = _usage(tmp_path)
    bad = []
    for k, (vg, sc, lds) in LIMITS.items():
        assert k in res, (k, sorted(res))
        u = res[k]
        if u["vgpr"] > vg or u["scratch"] > sc or u["lds"] > lds:
            bad.append((k, u, {"vgpr": vg, "scratch": sc, "lds": lds}))
    assert not bad, bad
