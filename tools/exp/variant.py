# Build a variant of the library from a patched COPY of csrc (experiments only; the product tree is not touched):
#   python tools/exp/variant.py NAME 'old1=>new1' 'old2=>new2' ...      -> ablx/libNAME.so
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, edits = sys.argv[1], sys.argv[2:]
dst = "/tmp/csrc_" + name
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(os.path.join(root, "pdb_eda_amd", "csrc"), dst)
inc = os.path.join(root, "include")
texts = {f: open(os.path.join(dst, f)).read() for f in os.listdir(dst)}
for e in edits:
    old, new = e.split("=>", 1)
    hit = False
    for f in texts:
        if old in texts[f]:
            texts[f] = texts[f].replace(old, new)
            hit = True
    assert hit, "pattern not found: " + old
for f, t in texts.items():
    open(os.path.join(dst, f), "w").write(t.replace('#include "../../include/pdbeda.h"', '#include "%s/pdbeda.h"' % inc))
os.makedirs(os.path.join(root, "ablx"), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function",
                       "-o", os.path.join(root, "ablx", "lib%s.so" % name), os.path.join(dst, "pdbeda_hip.hip")])
print("built ablx/lib%s.so" % name)
