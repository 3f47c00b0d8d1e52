#!/usr/bin/env python3
"""Trace one example operator on CPU tensors at a given size and report the generated library (nothing is
launched): source path, and -- with --isa -- a static count of the instructions of k_fwd by class.

    python tools/gen_kernel.py heat2d --Nt 256 --Nx 512 --Ny 512 --infer_k 1 --imposed stripe [--isa]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for sub in ("poisson", "heat", "velocity_from_tracer", "wave"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))


def isa_stats(lib):
    """Static instruction counts per kernel of the gfx950 code object embedded in `lib` + registers / LDS / scratch."""
    import tempfile

    llvm = "/opt/rocm/lib/llvm/bin/"
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        dis = subprocess.check_output([llvm + "llvm-objdump", "-d", co], text=True)
        notes = subprocess.check_output([llvm + "llvm-readelf", "--notes", co], text=True)
    meta = dict()
    for m in re.finditer(r"\.name:\s+(\w+)\n(.*?)(?=\n\s+- \.|\Z)", notes, re.S):
        pass
    cur, stats = None, collections.OrderedDict()
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            cur = stats.setdefault(m.group(1), collections.Counter())
            continue
        m = re.match(r"^\s+(\w+)", line)
        if cur is None or not m:
            continue
        op = m.group(1)
        cur["total"] += 1
        if op.startswith("v_"):
            cur["valu"] += 1
            if op.startswith("v_pk_"):
                cur["v_pk"] += 1
            if re.match(r"v_(exp|rcp|sqrt|rsq|log|sin|cos)_", op):
                cur["trans"] += 1
            if op.startswith("v_cndmask"):
                cur["cndmask"] += 1
            if op.startswith("v_mov") or op.startswith("v_accvgpr"):
                cur["mov"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
    regs = dict()
    for m in re.finditer(r"\.name:\s+(\w+)", notes):
        pass
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\w+)", block)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", block)
        sg = re.search(r"\.sgpr_count:\s+(\d+)", block)
        sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
        lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", block)
        ag = re.match(r"\s*(\d+)", block)
        if name:
            regs[name.group(1)] = dict(vgpr=int(vg.group(1)), agpr=int(ag.group(1)) if ag else 0, sgpr=int(sg.group(1)),
                                       scratch=int(sc.group(1)), lds=int(lds.group(1)))
    for k, c in stats.items():
        print("  {:14s} {}  {}".format(k, dict(c), regs.get(k, "")))


def main():
    import importlib

    argv = sys.argv[1:]
    isa = "--isa" in argv
    argv = [a for a in argv if a != "--isa"]
    modname, argv = argv[0], argv[1:]
    import odil_amd
    from odil_amd import runtime, stencil_jit

    runtime._mod = odil_amd.ModRocm(device="cpu")
    odil_amd.util.set_log_file(open(os.devnull, "w"))
    ex = importlib.import_module(modname)
    problem, state = ex.make_problem(ex.parse_args(argv))
    traced = stencil_jit.TracedOperator(problem, state)
    print("lib", traced.lib_path)
    src = traced.lib_path[:-3] + ".hip"
    if not os.path.exists(src):
        with open(src, "w") as f:
            f.write(traced.source)
    print("src", src)
    cg = traced.cg
    print("vw_fwd", cg.vw_fwd, "vw_gat", cg.vw_gat, "ncot", cg.ncot, "out_mode", cg.out_mode, "gathers", cg.gathers,
          "merged", cg.merged, "direct", cg.direct, "nparam", len(cg.pg_decl))
    if isa:
        isa_stats(traced.lib_path)


if __name__ == "__main__":
    main()
