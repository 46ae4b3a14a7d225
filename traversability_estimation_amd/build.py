"""Builds libtravgpu.so (the gfx950 HIP kernels + the C-ABI shim) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but travels with
the gpurun snapshot, so the GPU box uses the prebuilt file.  Every source is compiled to its own object
(in parallel, cached under _build/ by a CONTENT hash of the source, every header and the flags -- an unpack that resets
modification times cannot make a stale object look fresh; build_lib(force=True) compiles every object again,
build_lib(relink=True) only what is stale and then links) and the objects are linked into the library.

`python -m traversability_estimation_amd.build --lab` builds libtravgpu_lab.so (-DTE_LAB, objects under _build_lab/): the
same sources with their measurement switches (environment variables such as TE_NO_F4, TE_N3_BLOCKS_PER_CU) compiled in.
The shipped libtravgpu.so never reads the environment; tools/ load the lab library through TRAVGPU_LIB.
"""
import hashlib
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SOURCES = ["csrc/te_kernels.hip", "csrc/te_shim.hip", "csrc/te_transfer.hip", "csrc/te_paths_api.hip", "csrc/te_multi.hip", "csrc/te_stage.hip", "csrc/te_fast_step.hip", "csrc/te_step5.hip", "csrc/te_slide_normals.hip", "csrc/te_normals3.hip", "csrc/te_normals_small.hip", "csrc/te_footprint.hip", "csrc/te_footprint3.hip", "csrc/te_footprint4.hip", "csrc/te_footprint5.hip", "csrc/te_paths.hip", "csrc/te_gridmap_msg.hip", "csrc/te_polygon.hip", "csrc/te_trace.hip"]
# (every header of csrc/ and the public one; an object's key covers the ones it includes, see _deps)
HEADERS = ["csrc/te_internal.h", "csrc/te_march.h", "csrc/te_march5.h", "csrc/te_cell.h", "csrc/te_eig.h", "csrc/te_eig3.h", "csrc/te_geom.h", "csrc/te_msg.h", "csrc/te_ctx.h", "csrc/te_n3_plan.h", "csrc/te_hole_routing.h", "csrc/te_tie_triple.h", "../include/travgpu.h"]
LIB = os.path.join(_HERE, "libtravgpu.so")
LAB_LIB = os.path.join(_HERE, "libtravgpu_lab.so")
OBJDIR = os.path.join(_HERE, "_build")
LAB_OBJDIR = os.path.join(_HERE, "_build_lab")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-ldl"]
# per-source flags.  te_normals3: keep the ring reads as single ds_read_b64 -- a merged ds_read2_b64 halves the LDS rate
# (MI355X_MICROARCH.md, LDS table) and the kernel sits at 60 % LDS occupancy with them (4 % slower, same-box A/B)
_SINGLE_DS_READS = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt", "-mllvm", "-amdgpu-load-store-vectorizer=0"]
EXTRA_CFLAGS = {"csrc/te_normals3.hip": _SINGLE_DS_READS, "csrc/te_footprint3.hip": _SINGLE_DS_READS}
# sources compiled in several parts (-DTE_PARTS=n -DTE_PART=k, one object each): their shape-specialised kernels take
# minutes in one translation unit, and the parts compile side by side
PARTS = {"csrc/te_normals3.hip": 6, "csrc/te_footprint3.hip": 5, "csrc/te_footprint5.hip": 5}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm to build libtravgpu.so)")
    return exe


def _read(rel):
    with open(os.path.join(_HERE, rel), "rb") as f:
        return f.read()


_INCLUDE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)


def _deps(rel, seen=None):
    """The headers `rel` (a path relative to this directory) includes, transitively: csrc/ and ../include/."""
    seen = set() if seen is None else seen
    for name in _INCLUDE.findall(_read(rel).decode("utf-8", "replace")):
        for cand in (os.path.join(os.path.dirname(rel), name), os.path.join("csrc", name), os.path.join("..", "include", name)):
            cand = os.path.normpath(cand)
            if cand not in seen and os.path.exists(os.path.join(_HERE, cand)):
                seen.add(cand)
                _deps(cand, seen)
                break
    return seen


def _headers_digest():
    """(kept for callers that want one digest of every header: the library key)"""
    h = hashlib.sha256()
    for rel in HEADERS:
        h.update(rel.encode() + b"\0" + _read(rel))
    return h.digest()


def _key(unit, lab, headers=None):
    """Content key of one object: its source, the headers IT includes (transitively), the flags and the part it is --
    editing a header rebuilds the objects that see it, not the library."""
    src, part = unit
    h = hashlib.sha256()
    for rel in sorted(_deps(src)):
        h.update(rel.encode() + b"\0" + _read(rel))
    h.update(_read(src))
    h.update(repr((CFLAGS, EXTRA_CFLAGS.get(src, []), PARTS.get(src), part, bool(lab))).encode())
    return h.hexdigest()


def _fresh(unit, lab, headers=None):
    obj = _obj(*unit, lab=lab)
    try:
        with open(obj + ".key") as f:
            return os.path.exists(obj) and f.read().strip() == _key(unit, lab, headers)
    except OSError:
        return False


def _lib_key(lab, headers=None):
    headers = headers if headers is not None else _headers_digest()
    return hashlib.sha256(("".join(_key(u, lab, headers) for u in _units()) + repr(LDFLAGS)).encode()).hexdigest()


def stale(lab=False):
    lib = LAB_LIB if lab else LIB
    try:
        with open(lib + ".key") as f:
            return not os.path.exists(lib) or f.read().strip() != _lib_key(lab)
    except OSError:
        return True


def _units():
    """(source, part or None) for every object of the library."""
    return [(s, k) for s in SOURCES for k in (range(PARTS[s]) if s in PARTS else [None])]


def _obj(src, part=None, lab=False):
    base = os.path.basename(src)
    if part is not None:
        base = base[:-len(".hip")] + ".p%d.hip" % part
    return os.path.join(LAB_OBJDIR if lab else OBJDIR, base + ".o")


def _compile(unit, verbose, lab=False):
    src, part = unit
    defs = [] if part is None else ["-DTE_PARTS=%d" % PARTS[src], "-DTE_PART=%d" % part]
    if lab:
        defs.append("-DTE_LAB")
    out = _obj(src, part, lab)
    cmd = [hipcc()] + CFLAGS + EXTRA_CFLAGS.get(src, []) + defs + ["-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc"), "-c",
                               os.path.join(_HERE, src), "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    key = _key(unit, lab)  # (of the text that is about to be compiled)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    with open(out + ".key", "w") as f:
        f.write(key)


def build_lib(force=False, verbose=False, lab=False, relink=False):
    """force: every object is compiled again (about 6 minutes on 8 cores); relink: objects whose content key is stale
    are compiled, and the library is linked again even if nothing was (what __graft_entry__.build() asks
    for: on a fresh clone that IS a full build, on a warm tree it is the link check)."""
    lib = LAB_LIB if lab else LIB
    if not force and not relink and not stale(lab):
        return lib
    os.makedirs(LAB_OBJDIR if lab else OBJDIR, exist_ok=True)
    headers = _headers_digest()
    # force: every object is compiled again; otherwise only objects whose content key (source + headers + flags) differs
    # from the one they were compiled from
    todo = [u for u in _units() if force or not _fresh(u, lab, headers)]
    todo.sort(key=lambda u: u[0] not in PARTS)  # the long ones first
    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as pool:
        list(pool.map(lambda u: _compile(u, verbose, lab), todo))
    cmd = [hipcc()] + LDFLAGS + [_obj(*u, lab=lab) for u in _units()] + ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    with open(lib + ".key", "w") as f:
        f.write(_lib_key(lab))
    return lib


if __name__ == "__main__":
    import sys
    print(build_lib(force="--force" in sys.argv, relink=True, verbose=True, lab="--lab" in sys.argv))
