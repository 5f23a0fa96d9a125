python scripts/ab_two_libs_conv.py build_tmp/libfz_before_prio.so build_tmp/libfz_ch_prio3.so build_tmp/libfz_ch_prio1.so > $O/halo_prio_ab.txt 2>&1; cat $O/halo_prio_ab.txt
