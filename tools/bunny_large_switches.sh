export BUNNY_VERTS=1000000 BUNNY_CACHE=/tmp/bunny_1M.pkl
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z_bunny1M_switches.txt; : > $O
python $R/tools/bunny_time.py default >> $O || exit 1
for round in 1 2; do
python $R/tools/bunny_time.py default >> $O
SB_QUAD_LANES=256 python $R/tools/bunny_time.py quad_lanes_256 >> $O
SB_NO_WAVE_ITEMS=1 python $R/tools/bunny_time.py no_wave_items >> $O
SB_NO_WAVE_ITEMS=1 SB_QUAD_LANES=256 python $R/tools/bunny_time.py no_items_256 >> $O
TILE=384 python $R/tools/bunny_time.py tile384 >> $O
TILE=512 python $R/tools/bunny_time.py tile512 >> $O
TILE=192 python $R/tools/bunny_time.py tile192 >> $O
done
cat $O
