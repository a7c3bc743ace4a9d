"""Print registers / scratch / LDS of the kernels in libyolov3_hip.so whose mangled name contains the given text."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-yolov3_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_code_object as t  # noqa: E402

pat = sys.argv[1] if len(sys.argv) > 1 else ""
data = open(t.LIB, "rb").read()
for _, elf in t._code_objects(data):
    for k in t._kernels(elf):
        if pat in k[".name"]:
            print("%-100s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d wg %4d" % (
                k[".name"][:100], k[".vgpr_count"], k.get(".agpr_count", 0), k[".sgpr_count"],
                k[".private_segment_fixed_size"], k[".group_segment_fixed_size"], k[".max_flat_workgroup_size"]))
