"""dev: per-kernel resource summary (VGPRs, spills, scratch, LDS, SGPRs) of a hipcc -save-temps .s file.

    python tools_dev/kres.py file.s [substring ...]"""
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for blk in txt.split("  - .agpr_count:")[1:]:
    def g(key):
        m = re.search(r"\.%s:\s+(\S+)" % key, blk)
        return m.group(1) if m else "?"
    name = g("name")
    if pats and not any(p in name for p in pats):
        continue
    short = re.sub(r"^_ZN5waldoL?\d+", "", name)
    short = re.sub(r"EvPK.*$|EPK.*$", "", short)
    print(f"{short:60s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} "
          f"lds {g('group_segment_fixed_size'):>6s} sgpr {g('sgpr_count'):>4s}")
