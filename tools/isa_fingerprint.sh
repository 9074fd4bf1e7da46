#!/bin/bash
# tools/isa_fingerprint.sh <libhipbfv.so> -- one sha256 per device kernel of a built library (disassembly of the gfx950 code objects,
# addresses and symbol offsets stripped).  Two builds with the same fingerprints run the same instructions: the check behind
# "removed this knob / dead arm with byte-identical ISA" (no GPU needed).
set -e
LIB=$(readlink -f $1)
T=$(mktemp -d); cd $T
cp $LIB lib.so
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null
for co in lib.so.*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn --no-leading-addr $co
done | awk '
  /^[0-9a-f]* <.*>:$/ || /^<.*>:$/ { if (name != "") print name, cnt; name = $0; gsub(/^[0-9a-f]* /, "", name); cnt = 0; print "KERNEL " name; next }
  /^[ \t]+[a-z]/ { sub(/\/\/.*$/, ""); print }
' | python3 -c '
import sys, hashlib, re
cur, h, out = None, None, {}
for line in sys.stdin:
    if line.startswith("KERNEL "):
        cur = line[7:].strip(); h = hashlib.sha256(); out[cur] = [h, 0]; continue
    if cur is None or not line.startswith((" ", "\t")): continue
    t = re.sub(r"\s+", " ", line.strip())
    out[cur][0].update(t.encode()); out[cur][1] += 1
for k in sorted(out):
    if out[k][1] > 8: print(out[k][0].hexdigest()[:16], out[k][1], k)
'
rm -rf $T
