"""Identity of the kernel sources a measurement was taken on: sha256 over `ladcast_amd/csrc/*.{hip,inc,h}`, the Makefile and
`include/*.h` (sorted, contents only).  `tools/summarize_pmc.py` stamps it into `profiles/pmc_summary.json`; `bench.py` compares it
with the tree it runs from and flags a copied PMC figure as stale when the kernels have changed since the counter passes."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16() -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "ladcast_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ladcast_amd", "csrc", "*.inc"))
                   + glob.glob(os.path.join(ROOT, "ladcast_amd", "csrc", "*.h")) + [os.path.join(ROOT, "ladcast_amd", "csrc", "Makefile")]
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
