"""Diagnostic build of the HIP library under another name: python tools/diag/build_variant.py NAME -DFLAG ...
-> cvpr2021_vspw_implement_amd/lib/libvspw_hip_NAME.so (select it with VSPW_HIP_LIB=...).  Never the shipped library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "cvpr2021_vspw_implement_amd")


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    csrc = os.path.join(PKG, "csrc")
    objdir = os.path.join(PKG, "lib", "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            o = os.path.join(objdir, f[:-4] + ".o")
            objs.append(o)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags
                                          + ["-c", os.path.join(csrc, f), "-o", o]))
    assert all(p.wait() == 0 for p in procs)
    out = os.path.join(PKG, "lib", "libvspw_hip_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main()
