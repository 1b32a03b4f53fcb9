"""Randomised parity soak REPORT (GPU): the full protocol of tests/soak_lib.py on 36 batches (2304 solves), device vs
oracle next to the oracle-vs-itself control, plus the float-storage distribution.  Writes gpurun_out/<ROUND>_parity_soak.json (environment ROUND, default r06)
(copy into profiles/).  usage: python tests/soak/parity_soak.py [n_batches]"""
import json
import os
import sys

sys.path.insert(0, ".")
from tests import soak_lib  # noqa: E402
from tests.test_gpu_soak import DeviceStepper  # noqa: E402
import numpy as np  # noqa: E402
from direct_amd import solver  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 36
recs = soak_lib.soak(DeviceStepper, nb, control_seeds=(11, 12, 13))
n_out, n_cert, early = soak_lib.certificate(recs)


def plan(p0, p1, batch):
    s = solver.DdpSolver(batch.batch, batch.n_seg_max, batch.p_max, np.float32)
    r = s.plan(p0, p1, batch)
    s.close()
    return r


rep = dict(protocol="tests/soak_lib.py: %d batches x 32 problems x 2 phases, fp64, stepped per outer iteration" % nb,
           device_vs_oracle=soak_lib.summarise(recs, "impl"),
           control_oracle_vs_oracle_with_1ulp_inputs=[soak_lib.summarise(recs, ("control", i)) for i in range(3)],
           certificate=dict(solves_where_device_leaves_oracle=n_out, of_which_oracle_itself_is_sensitive=n_cert,
                            max_dev_after_first_iteration=early),
           float_storage_vs_fp64_oracle=soak_lib.float_storage_distribution(plan, nb))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/%s_parity_soak.json" % os.environ.get("ROUND", "r06"), "w"), indent=1)
print(json.dumps(rep, indent=1))
