# Run ON THE GPU BOX: the default bench line (live HBM traffic passes included), its duration, the fp16 operand bounds of the
# two weight recipes.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06g; mkdir -p $O
T0=$(date +%s); timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_stderr.txt; echo "bench.py wall: $(( $(date +%s) - T0 )) s"
cut -c1-3000 $O/bench_line.json; cp bench_detail.json $O/
python - <<'PY' 2>&1 | tail -5
import sys, types, torch
sys.path.insert(0, ".")
from faceformer_amd.models import SurfaceFormer_Parallel
from faceformer_amd.synth import make_state_dict, state_dict_spec
tok = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
for recipe in ("default", "gain4"):
    m = SurfaceFormer_Parallel(num_model=512, num_head=8, num_feedforward=1024, num_encoder_layers=6, num_decoder_layers=6, dropout=0.2, num_lines=256, max_face_length=37, token=tok)
    m.load_state_dict(make_state_dict(state_dict_spec("parallel", 256, 37, 512, 1024, 6, 6), recipe, 0))
    m = m.eval().cuda()
    e = m.engine()
    print(recipe, e.split_kind, {k: round(v, 2) for k, v in e.fp16_operand_bounds.items()})
PY
