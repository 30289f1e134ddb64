#!/usr/bin/env python3
"""Developer A/B builds of libiggt_hip.so (same ABI, one translation unit compiled with different flags) into
probes/lib_alt/<name>.so; load one with IGGT_HIP_LIB=<path> (iggt_official_amd/_C.py).  Not part of the product."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iggt_official_amd import build_ext  # noqa: E402

VARIANTS = {
    "attn_noslp": {"attention_v3.hip": ["-fno-slp-vectorize"]},
    "attn_nopin": {"attention_v3.hip": ["-DIGGT_ATTN_NO_PIN"]},
    "attn_nopin_noslp": {"attention_v3.hip": ["-DIGGT_ATTN_NO_PIN", "-fno-slp-vectorize"]},
    "hdb_stats": {"hdbscan.hip": ["-DIGGT_HDB_STATS"]},
    # round 5: the halo convolution without its epilogue (ablation: upper bound of what a cheaper epilogue could save)
    "conv_noepi": {"conv3x3_halo.hip": ["-DIGGT_CONV_NO_EPILOGUE"]},
    # round 6: ... without the lo half of the hi / lo split of the halo (ablation: upper bound of producer-written operand planes)
    "conv_nosplit": {"conv3x3_halo.hip": ["-DIGGT_CONV_NO_SPLIT"]},
    # estimated-shift instantiation of the static attention kernel (round 4; timed by probes/attn_est_ab.py)
    "est_default_sched": {"attention_v3_est.hip": []},
    "est_nodelta_maxilp": {"attention_v3_est.hip": ["-DIGGT_EST_NODELTA", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]},
    "est_nodelta": {"attention_v3_est.hip": ["-DIGGT_EST_NODELTA"]},
    **{f"est_dmax{d}": {"attention_v3_est.hip": [f"-DIGGT_EST_DELTA_MAX={d}.0f", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
       for d in (1, 2, 3, 4, 6, 8, 12)},
    # LLVM scheduling strategies on the three matrix-pipe kernels (timed by probes/sched_ab.py); iterative-ilp does not get
    # through attention_v3.hip (compiler error)
    **{f"sched_{n}": {f: fl for f in ("attention_v3.hip", "gemm_bf16_t256.hip", "conv3x3_halo.hip")}
       for n, fl in {"maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
                     "nopostra": ["-mllvm", "-enable-post-misched=false"],
                     "postbu": ["-mllvm", "-misched-postra-direction=bottomup"]}.items()},   # per-round tile counters, read by probes/hdbscan_profile.py
}


def main():
    build_ext.build(verbose=False)
    out_dir = os.path.join(ROOT, "probes", "lib_alt")
    os.makedirs(out_dir, exist_ok=True)
    names = sys.argv[1:] or [n for n in VARIANTS if not n.startswith(("sched_", "est_"))]
    for name in names:
        objs, procs = [], []
        for src in build_ext.sources():
            base = os.path.basename(src)
            extra = VARIANTS[name].get(base)
            if extra is None:
                objs.append(os.path.join(build_ext.LIB_DIR, "obj", base[:-4] + ".o"))
            else:
                obj = os.path.join(out_dir, f"{name}_{base[:-4]}.o")
                objs.append(obj)
                procs.append(subprocess.Popen([build_ext.HIPCC] + build_ext.FLAGS + extra + ["-c", src, "-o", obj]))
        for pr in procs:
            assert pr.wait() == 0
        lib = os.path.join(out_dir, name + ".so")
        subprocess.check_call([build_ext.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        print(lib)


if __name__ == "__main__":
    main()
