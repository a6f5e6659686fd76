#!/bin/bash
# round 6, fifteenth GPU run: rocprofv3 kernel trace + PMC passes of the ECDSA verifier's line (emitted 16-strand code, folded check):
# the counters behind roofline_r1cs.traffic / roofline_eval.traffic of config 5
cd "$GRAFT_REPO_ROOT" || exit 1
rm -f gpurun_out/traffic.json
bash tools/profile.sh r06u_ecdsa_verify_1024 ecdsa_verify:1024 --workload ecdsa_verify --in-flight 1 2>&1 | tail -40
