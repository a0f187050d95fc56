#!/bin/bash
# Regenerates ffwm_amd/miopen_db/ (run through gpurun from the repo root; ~15 GPU-minutes):
# one MIOpen find pass (torch.backends.cudnn.benchmark = True) over the convolutions of the bench workloads,
# written to a scratch user database that starts from the in-tree one; copy gpurun_out/miopen_db/*.txt back into
# ffwm_amd/miopen_db/ afterwards.  The database is MIOpen's own text format (solver timings per convolution
# configuration for gfx950 / 256 CUs): selection data only, no kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
DB=$R/gpurun_out/miopen_db
rm -rf $DB && mkdir -p $DB
cp $R/ffwm_amd/miopen_db/*.txt $DB/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$DB
for w in ${FFWM_FIND_WORKLOADS:-train flownet flowtrain}; do
    FFWM_MIOPEN_FIND=1 timeout 1300 python $R/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-kernels 2>/dev/null | tail -1 | cut -c1-160
done
rm -f $DB/*.time
ls -la $DB; du -sh $DB
for w in ${FFWM_FIND_WORKLOADS:-train flownet flowtrain}; do
    timeout 400 python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernels 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('immediate with find-db:', d['metric'], d['value'], d['ms_per_step'])"
done
