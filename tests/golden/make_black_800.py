"""Generator of the golden fixture black_800.sha256.

The reference commits two rendered outputs, output/final_scene.ppm and output/cornell_smoke.ppm.
Both are the same bytes: "P3\\n800 800\\n255\\n" followed by 640000 lines "0 0 0\\n"
(3 840 015 bytes).  The fixture stores only the sha256 of that text; this script re-creates the
text from its description and, when the reference checkout is present, checks it against the
two files."""
import hashlib
import os

text = b"P3\n800 800\n255\n" + b"0 0 0\n" * (800 * 800)
digest = hashlib.sha256(text).hexdigest()
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "black_800.sha256"), "w") as f:
    f.write(digest + "  P3 800x800 all-black (reference output/final_scene.ppm == output/cornell_smoke.ppm)\n")
for ref in ("/root/reference/output/final_scene.ppm", "/root/reference/output/cornell_smoke.ppm"):
    if os.path.exists(ref):
        assert hashlib.sha256(open(ref, "rb").read()).hexdigest() == digest, ref
        print("matches", ref)
print(digest, len(text))
