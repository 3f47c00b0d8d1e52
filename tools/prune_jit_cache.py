#!/usr/bin/env python3
"""Keeps odil_amd/_jit_cache to the generated kernels that are really loaded.

Every cache hit touches its library (`stencil_codegen._compile`: os.utime), so after a run the libraries with a
modification time newer than a marker are exactly the ones that were used (hits and fresh builds alike):

    python tools/prune_jit_cache.py mark                       # on the GPU box, before the run
    python -m pytest tests -m gpu -q; python bench.py; ...      # whatever should stay warm
    python tools/prune_jit_cache.py list > gpurun_out/jit_used.txt
    python tools/prune_jit_cache.py prune gpurun_out/jit_used.txt   # in the build container: everything else goes

Round 5: 963 libraries (51 MB; 715 at the end of round 4, most of them left behind by earlier versions of the generator:
any change of the emitted text changes every hash) -> 162 (9.4 MB) for the whole GPU suite, bench.py with all its
configurations, build() and smoke(); tools/prebuild_jit.py then adds the BASELINE configurations' multi-rank variants."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.environ.get("ODIL_JIT_CACHE", os.path.join(ROOT, "odil_amd", "_jit_cache"))
MARK = os.path.join(CACHE, ".prune_marker")


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "mark":
        open(MARK, "w").close()
    elif cmd == "list":
        t0 = os.path.getmtime(MARK)
        for f in sorted(os.listdir(CACHE)):
            if f.endswith(".so") and os.path.getmtime(os.path.join(CACHE, f)) >= t0:
                print(f)
    elif cmd == "prune":
        used = {l.strip() for l in open(sys.argv[2]) if l.strip()}
        removed = 0
        for f in os.listdir(CACHE):
            stem = f.rsplit(".", 1)[0]
            if (f.endswith(".so") and f not in used) or (f.endswith(".hip") and stem + ".so" not in used):
                os.unlink(os.path.join(CACHE, f))
                removed += f.endswith(".so")
        print("removed {} libraries, {} left".format(removed, sum(f.endswith(".so") for f in os.listdir(CACHE))))
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
