"""Extracts the lines the reference's own harness compares (tests/extra/staghmc_sh/run:43-44:
MEASploop | MEASplaq | MEASpbp | (Begin|End|Reversed) H, time stamps stripped) plus the solver statistics
from its golden logs tests/extra/staghmc_sh/ref.{0,1,2} into tests/golden/staghmc_sh/ref.N.check.
Run in the build container (needs /root/reference); the .check files are committed data."""
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/extra/staghmc_sh"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "staghmc_sh")
keep = re.compile(r"^MEASploop|^MEASplaq|^MEASpbp|(Begin|End|Reversed) H:|^stagSolve:|^Solver\[|^  [AF] m=|^(ACCEPT|REJECT)")
for n in (0, 1, 2):
    with open(os.path.join(REF, "ref.%d" % n)) as f, open(os.path.join(OUT, "ref.%d.check" % n), "w") as o:
        for line in f:
            if keep.search(line):
                o.write(re.sub(r"^\[[^]]*\] *", "", line))
