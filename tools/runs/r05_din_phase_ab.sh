for v in nopf2 pf2; do echo "== $v"; REC_DIN_FWD_VARIANT=$v python tools/din_phase_probe.py 4096 100 | tail -11; done
