import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import dbg_patch_stress as S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
os.system('rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i "unique\\|serial" | head -3')
for shape in ((8, 32, 32, 64, 128, 5, 2), (8, 16, 16, 128, 256, 5, 2), (64, 8, 8, 400, 800, 5, 2)):
    print('shape', shape, flush=True)
    S.run(*shape, reps)
