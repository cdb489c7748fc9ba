#!/bin/bash
# Register / LDS / spill figures of every kernel of an object file's gfx950 code object:
#   tools/kernel_meta.sh zk-saas_amd/csrc/msm_bn254_g1.o [name filter]       (KEEP=dir keeps the code object there)
O=$1; F=${2:-.}
T=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin $O $T/copy.o
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fb.bin --output=$T/dev.co --unbundle
[ -n "$KEEP" ] && cp $T/dev.co $KEEP/$(basename $O .o).co
$B/llvm-readelf --notes $T/dev.co | python3 -c '
import sys,re,subprocess
txt=sys.stdin.read()
rows=[]
for blk in txt.split("  - .agpr_count:")[1:]:
    g=lambda k:(re.search(r"\.%s:\s+(\S+)"%k,blk) or [None,"?"])[1]
    rows.append((g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("name")))
names=subprocess.run(["c++filt"],input="\n".join(r[-1] for r in rows),capture_output=True,text=True).stdout.split("\n")
for r,n in zip(rows,names):
    print("vgpr %-4s agpr %-4s sgpr %-4s spill %-4s lds %-7s scratch %-6s %s"%(r[0],r[1],r[2],r[3],r[4],r[5],n[:140]))
' | grep -E "$F"
rm -rf $T
