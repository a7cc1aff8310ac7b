bash tools/r5_check.sh "track_golden or replicated or full_length or split_variants or random_scenes"
SGX_LIB=$PWD/softgnss-python_amd/lib/variants/libsgx_cnt.so python tools/step_profile.py 37000 2>&1 | grep "t3 count\|^step" | sort -u | awk '/unit  0 wave [02]|unit 18|unit 19|unit  9 wave 0|^step/'
bash tools/trk_ab.sh 2 default old
