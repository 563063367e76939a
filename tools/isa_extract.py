"""Print one kernel's ISA from a hipcc -S listing: python tools/isa_extract.py file.s <substring of the mangled name> [from-nth-barrier before after]."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
body = lines[start:end]
if len(sys.argv) > 3:
    nth, before, after = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    bars = [i for i, l in enumerate(body) if re.search(r"\ts_barrier", l)]
    c = bars[nth]
    body = body[max(0, c - before):c + after]
print("\n".join(l[:110] for l in body))
