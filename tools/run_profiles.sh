#!/bin/bash
# tools/run_profiles.sh ROUNDTAG — on the GPU box: bench + rocprofv3 stats + PMC passes for BASELINE configs C5, C2, C3, C4, C1
T=${1:-r02}
tools/gpu_profile.sh ${T}_C5 > gpurun_out/${T}_C5.log 2>&1
tools/gpu_profile.sh ${T}_C2 --scene random_spheres --nx 1200 --ny 800 --spp 500 > gpurun_out/${T}_C2.log 2>&1
tools/gpu_profile.sh ${T}_C3 --scene cornell_box --nx 800 --ny 800 --spp 1000 > gpurun_out/${T}_C3.log 2>&1
tools/gpu_profile.sh ${T}_C4 --scene cornell_smoke --nx 800 --ny 800 --spp 1000 > gpurun_out/${T}_C4.log 2>&1
tools/gpu_profile.sh ${T}_C1 --scene two_spheres --nx 400 --ny 225 --spp 100 > gpurun_out/${T}_C1.log 2>&1
for f in gpurun_out/${T}_C*.log; do tail -n 2 $f; done
