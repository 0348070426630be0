# chips-per-tile sweep of the persistent MEDA observation kernel (MEDA_VEC_OBS_TILE knob), both observation versions
for E in 262144 65536; do for t in 4 5 6 7 8 9 10 11 12 13 14; do
  echo "== E $E chips/tile $t: v0, v0_2"; MEDA_VEC_OBS_TILE=$t python tools/ab_meda_obs.py $E 0 main; MEDA_VEC_OBS_TILE=$t python tools/ab_meda_obs.py $E 2 main; done; done
