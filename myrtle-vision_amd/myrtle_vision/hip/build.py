"""Build libmyrtle_vision_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m myrtle_vision.hip.build [--force]

The library is built IN TREE (``myrtle-vision_amd/lib/``) so it travels with the source snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # myrtle-vision_amd/
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_DIR = os.path.join(PKG_ROOT, "lib")
LIB_PATH = os.environ.get("MV_LIB_PATH") or os.path.join(LIB_DIR, "libmyrtle_vision_hip.so")   # MV_LIB_PATH: diagnostic builds
INCLUDE = os.path.join(os.path.dirname(PKG_ROOT), "include")
SOURCES = ["layernorm.hip", "gemm_bf16.hip", "gemm_f32.hip", "attention.hip", "attention_f32.hip", "elementwise.hip", "seg_tail.hip",
           "image_prep.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per file.  attention.hip: its softmax arithmetic sits between MFMAs, where a packed f32 operation (what SLP vectorisation makes of
# adjacent scalar ones) costs more issue time than the two it replaces (MI355X_MICROARCH.md, vector-instruction issue cost)
EXTRA_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _torch_lib_dir():
    """Directory of the HIP runtime PyTorch-ROCm ships (found without importing torch), or None."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None
    except Exception:
        return None


def _link_command(hipcc, objs):
    """ONE HIP runtime per process.  PyTorch-ROCm bundles its own runtime (file ``torch/lib/libamdhip64.so``, soname
    ``libamdhip64.so.7``) and requests it by FILE name; this library requests ``libamdhip64.so.7``.  If torch is in
    the process first, the dynamic loader satisfies our request with torch's already-loaded instance (soname match):
    one runtime, shared streams and device state -- which is why ``lib.py`` imports torch before ``CDLL``.  Loaded the
    other way round, torch's file-name request would NOT match the /opt/rocm instance and a second runtime would
    start ("no ROCm-capable device" on the first launch, observed).  Linking with g++ against torch's copy (device
    code is already embedded in the objects) only makes the RUNPATH prefer that copy; a non-torch host falls through
    to /opt/rocm/lib."""
    tl = _torch_lib_dir()
    if tl is None:
        return [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    return ["g++", "-shared", "-fPIC", "-o", LIB_PATH] + objs + [
        f"-L{tl}", "-l:libamdhip64.so", f"-Wl,-rpath,{tl}:{rocm}/lib", "-Wl,--enable-new-dtags"]


def _digest():
    h = hashlib.sha256()
    # every file of csrc/ itself (csrc/diag/ holds retired kernels kept as source for the record: not built, not digested)
    files = [f for f in sorted(os.listdir(CSRC)) if os.path.isfile(os.path.join(CSRC, f))]
    for f in files + ["../../include/myrtle_vision_hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    h.update(str(_torch_lib_dir()).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp = os.path.join(LIB_DIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB_PATH
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run(_link_command(hipcc, objs), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
