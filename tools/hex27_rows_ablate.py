"""TIMING-ONLY ablations of k_hex27_rows_gq on a fully distorted N^3 mesh (bits 12-14 of mfem_debug_set_hex27): where its time goes.  usage: hex27_rows_ablate.py [N]"""
import subprocess, sys
for ab, name in ((0, "as is"), (1, "no arithmetic (+ no additions)"), (2, "no LDS additions"), (4, "no G_q loads"), (7, "row stores + barriers only")):
    out = subprocess.run([sys.executable, __file__.replace("hex27_rows_ablate", "hex27_rows_time"), sys.argv[1] if len(sys.argv) > 1 else "128", str(ab << 12)],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    print(f"{name:40s} {out}", flush=True)
