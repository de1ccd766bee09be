import sys, os, re, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_lib as O
from skirt9_amd.host import Simulation
from skirt9_amd.engine import Engine
base = open("tests/ski/cfg3small.ski").read()
inst = re.findall(r"    <F\w+Instrument instrumentName=\"i\d\".*?/>\n", base)
variants = {"all": inst, "i0": inst[:1], "i0+i1": inst[:2], "i2": [inst[2].replace('"i2"', '"i0"')]}
for name, sel in variants.items():
    txt = base
    for s in inst: txt = txt.replace(s, "")
    txt = txt.replace("   <instruments type=\"Instrument\">\n", "   <instruments type=\"Instrument\">\n" + "".join(sel))
    path = os.path.join(tempfile.gettempdir(), "dbg_%s.ski" % name.replace("+", "_"))
    open(path, "w").write(txt)
    n = 5000
    sim = Simulation(path, num_packets=n).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, n, 12345)
    gpu = eng.download()
    ref, c = O.run_primary(sim, 0, n, O.RNG_PHILOX, seed=12345)
    g = eng.counters()
    print(name, "gpu", {k: g[k] for k in ("paths", "cell_visits", "rewalk_visits", "scatterings", "detector_updates")},
          "oracle", c.paths, c.cell_visits, c.scatterings, c.detector_updates, "max rel frame diff",
          float(np.max(np.abs(gpu - ref)) / np.max(np.abs(ref))))
