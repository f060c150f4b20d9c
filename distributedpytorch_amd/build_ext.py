"""In-tree build of the gfx950 extension ``distributedpytorch_amd/_C*.so``.

hipcc is driven directly (no hipify pass, no JIT cache): kernel ``.hip`` files
include no torch headers and compile in seconds; only the two binding files see
``torch/extension.h``.  Objects are cached under ``build/`` keyed by source hash,
so an unchanged file is never recompiled.  The ``.so`` lands next to the package
so it travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
BUILD = PKG.parent / "build" / "dpa_ext"
ARCH = "gfx950"
MODULE = "_C"


def so_path() -> Path:
    return PKG / f"{MODULE}{sysconfig.get_config_var('EXT_SUFFIX')}"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce
    inc = []
    for p in ce.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"]]
    lib_dir = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    return inc, lib_dir, abi


def _sources():
    hip = sorted(CSRC.glob("*.hip"))
    cpp = sorted(CSRC.glob("*.cpp"))
    return hip, cpp


def _digest(path: Path, extra: str) -> str:
    h = hashlib.sha1()
    h.update(path.read_bytes())
    for hdr in sorted(CSRC.glob("*.h")):
        h.update(hdr.read_bytes())
    h.update(extra.encode())
    return h.hexdigest()[:16]


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed:\n" + " ".join(map(str, cmd)) + "\n" + r.stdout)
    return r.stdout


def _check_loads(so: Path) -> None:
    """dlopen the fresh library in a child interpreter: an unresolved symbol (e.g. a kernel whose host
    stub the compiler dropped) must fail the build here, not the first GPU run."""
    code = ("import importlib.util, sys, torch\n"
            f"spec = importlib.util.spec_from_file_location('distributedpytorch_amd.{MODULE}', r'{so}')\n"
            "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        so.unlink(missing_ok=True)
        raise RuntimeError(f"{so.name} was linked but does not load:\n{r.stdout[-2000:]}")


def build(verbose: bool = True, jobs: int = 8) -> Path:
    BUILD.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()
    inc, lib_dir, abi = _torch_flags()
    hip_srcs, cpp_srcs = _sources()
    have_attention = (CSRC / "attention.hip").exists()
    common = ["-O3", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"]
    if have_attention:
        common.append("-DDPA_HAVE_ATTENTION=1")
    kernel_flags = common + [f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result"]
    bind_flags = common + inc + [f"-DTORCH_EXTENSION_NAME={MODULE}", "-DTORCH_API_INCLUDE_EXTENSION_H",
                                 "-Wno-deprecated-declarations", "-I/opt/rocm/include"]
    tasks = []
    objs = []
    for src in hip_srcs:
        tag = _digest(src, " ".join(kernel_flags))
        obj = BUILD / f"{src.stem}.{tag}.o"
        objs.append(obj)
        if not obj.exists():
            tasks.append((src, [hipcc, *kernel_flags, "-c", str(src), "-o", str(obj)]))
    for src in cpp_srcs:
        tag = _digest(src, " ".join(bind_flags))
        obj = BUILD / f"{src.stem}.{tag}.o"
        objs.append(obj)
        if not obj.exists():
            tasks.append((src, [hipcc, "-x", "c++", *bind_flags, "-c", str(src), "-o", str(obj)]))

    def compile_one(t):
        src, cmd = t
        if verbose:
            print(f"[build_ext] compiling {src.name}", flush=True)
        _run(cmd)

    if tasks:
        with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(tasks)))) as ex:
            list(ex.map(compile_one, tasks))
    out = so_path()
    stamp = BUILD / "link.stamp"
    key = " ".join(str(o) for o in objs)
    if tasks or not out.exists() or not stamp.exists() or stamp.read_text() != key:
        if verbose:
            print(f"[build_ext] linking {out.name}", flush=True)
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *map(str, objs), "-o", str(out),
              f"-L{lib_dir}", "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
              "-lamdhip64", "-lrccl", f"-Wl,-rpath,{lib_dir}"])   # the RCCL torch itself loads (one RCCL per process)
        _check_loads(out)
        stamp.write_text(key)
        # drop objects of older source versions
        keep = {o.name for o in objs}
        for old in BUILD.glob("*.o"):
            if old.name not in keep:
                old.unlink()
    return out


if __name__ == "__main__":
    p = build(verbose=True)
    print(p)
    sys.exit(0)
