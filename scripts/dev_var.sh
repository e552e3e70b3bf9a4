for v in 1e-9_4 1e-10_4 1e-11_4; do
  cp bilevel-gait-gen_amd/libsrbm_rti_v_$v.so bilevel-gait-gen_amd/libsrbm_rti.so
  echo "variant $v"; python scripts/dev_inst.py 1e-13 2>&1 | tail -11 | cut -c1-420 | head -3; python scripts/dev_time.py a1_configuration 256 10 | tail -1
done
