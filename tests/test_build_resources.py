"""What the compiler made of the kernels, read from the built library (no GPU needed).

Builds of the wavefront kernels in which the shading kernel spilled registers to scratch memory have ended with
illegal memory accesses on the GPU (scenes with sphere lights) where the same source without spills passes every
test; so have builds with loads through generic ("flat") pointers.  The kernels below must therefore come out of the
compiler without scratch memory, and the wavefront renderer without flat instructions: a change that makes the
compiler spill in one of them is caught here, before the library goes near a GPU."""
import os
import re
import shutil
import subprocess

import pytest

import path_tracing_amd as pt

LLVM = "/opt/rocm/lib/llvm/bin"

# kernels that run in every render and must not use scratch memory (name prefixes after demangling by eye)
NO_SCRATCH = ["shadeHitsKernel", "classifyKernel", "generateKernel", "resolveKernel", "finishPathsKernel", "walkLinksKernel",
              "traceRetryKernel", "quadMarkKernel", "quadRowsKernel"]
# template arguments <WIDTH, DYN, STATS> / <DYN, LDS, STATS, ...>: the variants for scenes without moving surfaces, not instrumented
NO_SCRATCH_STATIC = [r"traceOrderedKernelILi[24]ELb0ELb0", r"traceKernelILb0ELb[01]ELb0"]


def code_objects(tmp_path):
    lib = str(tmp_path / "libptamd.so")
    shutil.copy(pt.LIB_PATH, lib)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True, cwd=str(tmp_path))
    return [str(tmp_path / f) for f in sorted(os.listdir(str(tmp_path))) if "amdgcn" in f]


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="no LLVM tools")
def test_kernels_that_must_not_spill(tmp_path):
    objs = code_objects(tmp_path)
    assert objs
    kernels = {}
    for o in objs:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.search(r"\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name:
                kernels[name] = int(m.group(1))
    assert len(kernels) > 40
    seen = set()
    for name, scratch in kernels.items():
        for want in NO_SCRATCH:
            if re.search(r"\d+%s" % want, name):
                seen.add(want)
                assert scratch == 0, (name, scratch)
        for want in NO_SCRATCH_STATIC:
            if re.search(want, name):
                seen.add(want)
                assert scratch == 0, (name, scratch)
    assert seen == set(NO_SCRATCH) | set(NO_SCRATCH_STATIC), sorted((set(NO_SCRATCH) | set(NO_SCRATCH_STATIC)) - seen)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no LLVM tools")
def test_wavefront_kernels_have_no_flat_instructions(tmp_path):
    flat = {}
    for o in code_objects(tmp_path):
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", o], check=True, capture_output=True, text=True).stdout
        if "shadeHitsKernel" not in text:
            continue
        current = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                current = m.group(1)
            elif re.search(r"\sflat_(load|store|atomic)", line) and current:
                flat[current] = flat.get(current, 0) + 1
    assert not flat, flat
