"""Print registers / spills / scratch of every kernel in librarc_hip.so (reads the embedded gfx950 code objects; no GPU).
usage: python tools/kernel_resources.py [substring]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import codeobj  # noqa: E402

if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for name, v in sorted(codeobj.kernel_resources().items()):
        if pat and pat not in name:
            continue
        print(f"{v.get('vgpr_count', 0):5d} {v.get('agpr_count', 0):5d} {v.get('sgpr_count', 0):5d} "
              f"{v.get('vgpr_spill_count', 0):6d} {v.get('sgpr_spill_count', 0):6d} "
              f"{v.get('private_segment_fixed_size', 0):7d} {v.get('group_segment_fixed_size', 0):7d}  {codeobj.demangle(name)[:110]}")
