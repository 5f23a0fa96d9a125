FZ_TIMELINE_BRIEF=1 timeout 300 ./build_tmp/igemm_timeline shortk -2 2>&1 | grep "wall clock" | sed 's/wave 0 of workgroup 0: //' | cut -c1-40,150-260 > $O/with_bias.txt
FZ_TIMELINE_NOBIAS=1 FZ_TIMELINE_BRIEF=1 timeout 300 ./build_tmp/igemm_timeline shortk -2 2>&1 | grep "wall clock" | sed 's/wave 0 of workgroup 0: //' | cut -c1-40,150-260 > $O/no_bias.txt
paste -d'|' $O/with_bias.txt $O/no_bias.txt | cut -c1-300
