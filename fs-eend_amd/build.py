"""Build libeend_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery:
the library is a plain C-ABI shared object loaded with ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "ffn.hip", "ffn_stream.hip", "convert_f32.hip", "convert_rows.hip", "encin.hip", "conv_stream.hip", "proj.hip", "attn.hip", "attn_full.hip", "attn_stream.hip", "spk_stream.hip", "embloss.hip", "postproc.hip", "feature.hip", "pit.hip", "misc.hip", "retention.hip", "retention_full.hip", "ret_stream.hip", "stream.hip", "skinny.hip", "gemm_f32.hip", "api.hip",
           # training step: backward kernels, optimiser
           "wgrad.hip", "ffn_train_stream.hip", "proj_stream.hip", "gemm_acc_stream.hip", "attn_bwd.hip", "attn_bwd_fused.hip", "train_rows.hip", "embloss_bwd.hip", "optim.hip", "api_train.hip",
           # LS-EEND training step
           "ls_train.hip", "retention_bwd.hip"]
LIB = os.path.join(CSRC, "libeend_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    missing = [s for s in srcs if not os.path.exists(s)]
    if missing:
        raise FileNotFoundError(f"HIP sources listed in build.SOURCES are missing: {missing}")
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "eend_hip.h"))
    objs = [s[:-4] + ".o" for s in srcs]
    hipcc = _hipcc()

    def cc(pair):
        src, obj = pair
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
