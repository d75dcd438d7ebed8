mkdir -p gpurun_out/r3q
timeout -k 10 600 python -m pytest tests/test_gpu_voxel_conv.py tests/test_gpu_backbone_pipeline.py tests/test_gpu_backbone_res.py -x -q -m gpu > gpurun_out/r3q/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 gpurun_out/r3q/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/bench_backbone.py --batch 20 --mfma 1 > gpurun_out/r3q/b20.json 2> gpurun_out/r3q/b20.err; echo "b20 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r3q/b20.json').read().strip().splitlines()[-1]); print(d['ms_per_frame'], d['roofline']['frac'], d.get('ms_per_forward'))"
timeout -k 10 300 python tools/bench_backbone.py --batch 1 --mfma 1 > gpurun_out/r3q/b1.json 2> gpurun_out/r3q/b1.err; echo "b1 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r3q/b1.json').read().strip().splitlines()[-1]); print(d['ms_per_frame'], d['roofline']['frac'], d.get('ms_per_forward'))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bb
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bb -o ks -- python3 $GRAFT_REPO_ROOT/tools/bench_backbone.py --batch 20 --mfma 1 --reps 6 > $GRAFT_REPO_ROOT/gpurun_out/r3q/b20_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3q/b20_prof.err; echo "prof rc=$?"
cp /tmp/prof_bb/ks_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r3q/r03_backbone_b20_kernel_stats.csv
head -25 /tmp/prof_bb/ks_kernel_stats.csv | cut -c1-150
rm -rf /tmp/prof_bb1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bb1 -o ks -- python3 $GRAFT_REPO_ROOT/tools/bench_backbone.py --batch 1 --mfma 1 --reps 20 > $GRAFT_REPO_ROOT/gpurun_out/r3q/b1_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3q/b1_prof.err; echo "prof1 rc=$?"
cp /tmp/prof_bb1/ks_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r3q/r03_backbone_b1_kernel_stats.csv
