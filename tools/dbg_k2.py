import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
from mvil_fusion_amd.abi import Window
import oracle_lib
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg_chain.py")).read().split("cases =")[0].split("be = lib.open_vilsolve()")[1])
orc = oracle_lib.open_oracle(); be = lib.open_vilsolve()
for it in (1, 2, 3, 6):
    wg, wo = k2(), k2()
    opts = abi.default_options(max_iterations=it)
    sg = be.solve(wg, opts); so = orc.solve(wo, opts)
    print("it", it, "GPU iters", sg.iterations, "succ", sg.successful_steps, "term", sg.termination, "cost trace", [round(sg.cost_trace[i], 3) for i in range(sg.iterations)], "radius", [sg.radius_trace[i] for i in range(sg.iterations)])
    print("      ORC iters", so.iterations, "succ", so.successful_steps, "term", so.termination, "cost trace", [round(so.cost_trace[i], 3) for i in range(so.iterations)], "radius", [so.radius_trace[i] for i in range(so.iterations)])
    print("      final", sg.final_cost, so.final_cost, "dlam", np.abs(wg.inv_depth - wo.inv_depth).max(), "dpose", np.abs(wg.pose - wo.pose).max())
