#!/usr/bin/env python
"""Instruction mix of one kernel in a hipcc --save-temps .s file: isa_stats.py FILE.s SUBSTRING_OF_MANGLED_NAME"""
import re
import sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l and l.rstrip().split(":")[0].count(" ") == 0)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = "\n".join(lines[start:end])
def cnt(p):
    return len(re.findall(p, body))
print(lines[start][:60], "instructions:", cnt(r"\n\s+[vsdg][a-z_0-9]+ "))
print(" s_barrier %d  s_waitcnt %d  vmcnt(0) %d  lgkmcnt(0) %d" % (cnt("s_barrier"), cnt("s_waitcnt"), cnt(r"vmcnt\(0\)"), cnt(r"lgkmcnt\(0\)")))
print(" ds_write_b128 %d ds_write_b64 %d ds_write2 %d | ds_read_b128 %d ds_read_b64 %d ds_read2 %d" % (
    cnt("ds_write_b128"), cnt(r"ds_write_b64"), cnt("ds_write2"), cnt("ds_read_b128"), cnt(r"ds_read_b64"), cnt("ds_read2")))
print(" v_fma_f64 %d v_mul_f64 %d v_add_f64 %d | all VALU %d | int/move VALU %d" % (
    cnt("v_fma_f64"), cnt("v_mul_f64"), cnt("v_add_f64"), cnt(r"\n\s+v_"), cnt(r"\n\s+v_") - cnt(r"\n\s+v_[a-z]+_f64")))
print(" global_load %d global_store %d scratch %d v_accvgpr %d" % (cnt("global_load"), cnt("global_store"), cnt("scratch_"), cnt("v_accvgpr")))
