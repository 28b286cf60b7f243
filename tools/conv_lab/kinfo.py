import re, sys
s = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/conv_lab_tmp/conv_lab-hip-amdgcn-amd-amdhsa-gfx950.s").read()
for blk in s.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = g("name").group(1)
    if "conv" not in name: continue
    print(name[:48], "vgpr", g("vgpr_count").group(1), "agpr", blk.split()[0], "sgpr", g("sgpr_count").group(1), "scratch", g("private_segment_fixed_size").group(1), "lds", g("group_segment_fixed_size").group(1))
