"""Regenerates tests/golden/libc_rand_seed1.json from the C library of the machine it runs on."""
import ctypes, json, os
libc = ctypes.CDLL("libc.so.6")
libc.srand(1)
vals = [libc.rand() for _ in range(16)]
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libc_rand_seed1.json")
doc = json.load(open(path))
doc["values"] = vals
json.dump(doc, open(path, "w"), indent=2)
print(vals)
