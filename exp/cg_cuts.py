"""Lab builds of k_cg_slab with phases cut away (the method of DESIGN.md §4.8: results of a cut build are wrong, only durations / counters are read).  Patches a COPY of the tree
(the product sources carry no cut points) and builds exp/libmor_cgcut_<name>.so for: load (cells + rows staged, nothing else), a1 (+ enumeration, queued pairs dropped), a (+ pair
decisions A2), b1 (+ thread-level point tests), slab (whole slab phase, no merge), full.   usage (here, no GPU needed): python exp/cg_cuts.py"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamicslamtool_amd import build
CUTS = {"load": (1, 0), "a1": (4, 0), "a": (2, 0), "b1": (3, 0), "slab": (99, 0), "full": (99, 99)}


def patch(text):
    def ins(anchor, code, before=True):
        assert anchor in text_[0], anchor[:60]
        text_[0] = text_[0].replace(anchor, code + anchor if before else anchor + code, 1)
    text_ = [text]
    ins("  if (!d.cg_fused) return;\n  if (!stream_last_block(", "#if CGF_CUT == 0\n  return;\n#endif\n")
    ins("  cgs_hooks<LDS, BOXL, RT>(d, G, so + c0, n_own, n_loc, L, start, rows, rsub, r0, nlrows, par, sp, ovf, l_list, l_queue, l_wcnt, l_n2);\n", "#if CGS_CUT == 1\n  return;\n#endif\n")
    ins("  if (lane == 0) l_wcnt[w] = min(wcount, cgs_wlist_cap<LDS>());\n  __threadfence_block();\n  __syncthreads();\n", "#if CGS_CUT == 2 || CGS_CUT == 4\n  return;\n#endif\n", before=False)
    ins("  // ---- B2: one wave per pair left over\n", "#if CGS_CUT == 3\n  return;\n#endif\n")
    ins("    for (int q0 = 0; q0 < qn; q0 += 64) {", "#if CGS_CUT == 4\n    qn = 0;\n#endif\n")
    return text_[0]


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as td:
        shutil.copytree(os.path.join(ROOT, "dynamicslamtool_amd", "csrc"), os.path.join(td, "dynamicslamtool_amd", "csrc"), ignore=shutil.ignore_patterns("*.so", "mor_replay*"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(td, "include"))
        f = os.path.join(td, "dynamicslamtool_amd", "csrc", "kernels_cellgraph.h")
        patched = patch(open(f).read())
        open(f, "w").write(patched)
        procs = []
        for name, (cs, cf) in CUTS.items():
            out = os.path.join(ROOT, "exp", "libmor_cgcut_%s.so" % name)
            cmd = [build._hipcc()] + build.HIP_FLAGS + ["-DCGS_CUT=%d" % cs, "-DCGF_CUT=%d" % cf, '-DMOR_SRC_HASH_STR="MOR_SRC_HASH=cgcut%019d"' % 0, "-x", "hip"] + [os.path.join(td, "dynamicslamtool_amd", "csrc", x) for x in build.HIP_SOURCES] + ["-o", out]
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)))
        for name, p in procs:
            err = p.communicate()[1]
            print(name, "ok" if p.returncode == 0 else "FAILED\n" + err[-800:])
