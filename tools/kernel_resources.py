#!/usr/bin/env python3
"""Register, LDS and scratch use of every kernel in the built library, from the code object's metadata:

    python tools/kernel_resources.py [niwqg_amd/libniwqg_amd.so] [--scratch] [filter-substring ...]

(llvm-objdump --offloading extracts the gfx950 code object, llvm-readelf --notes prints the kernel descriptors.)  Runs anywhere
the ROCm LLVM tools are; no GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    lib = os.path.abspath(lib)
    with tempfile.TemporaryDirectory() as tmp:
        link = os.path.join(tmp, "lib.so")
        os.symlink(lib, link)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", link], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            raise SystemExit("no gfx950 code object in " + lib)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co[0])], check=True,
                               capture_output=True, text=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk)
        name = get("name").group(1)
        out.append(dict(name=name, vgpr=int(get("vgpr_count").group(1)), sgpr=int(get("sgpr_count").group(1)),
                        scratch=int(get("private_segment_fixed_size").group(1)), lds_static=int(get("group_segment_fixed_size").group(1)),
                        vgpr_spill=int(get("vgpr_spill_count").group(1)) if get("vgpr_spill_count") else 0,
                        sgpr_spill=int(get("sgpr_spill_count").group(1)) if get("sgpr_spill_count") else 0))
    names = subprocess.run(["c++filt"] + [k["name"] for k in out], capture_output=True, text=True).stdout.strip().split("\n")
    for k, n in zip(out, names):
        k["demangled"] = re.sub(r"\(.*", "", n.replace("void ", ""))
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    only_scratch = "--scratch" in args              # list only the kernels that use scratch
    args = [a for a in args if a != "--scratch"]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                            "niwqg_amd", "libniwqg_amd.so")
    ks = sorted(kernels(lib), key=lambda k: k["demangled"])
    print("%-64s %5s %5s %8s %11s %11s" % ("kernel", "VGPR", "SGPR", "scratch", "VGPR spills", "SGPR spills"))
    for k in ks:
        if (args and not any(a in k["demangled"] for a in args)) or (only_scratch and not k["scratch"]):
            continue
        print("%-64s %5d %5d %6d B %11d %11d" % (k["demangled"][:64], k["vgpr"], k["sgpr"], k["scratch"], k["vgpr_spill"], k["sgpr_spill"]))
    bad = [k for k in ks if k["scratch"]]
    print("%d kernels, %d with scratch" % (len(ks), len(bad)))
