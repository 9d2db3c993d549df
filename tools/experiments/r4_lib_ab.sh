# same-box A/B of two builds of the library: the in-tree one against variants/libdsp_B.so (DSP_AMD_LIB); usage: r4_lib_ab.sh [batch ...]
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/variants/libdsp_B.so
cat > /tmp/dump.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
from oracle import forward_np as onp
cfg = onp.OracleConfig(); w = onp.make_weights(cfg, 91, 2.0)
m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0, cfg.hidden_size, cfg.vocab_size,
                cfg.embedding_size, cfg.is_base, cfg.is_signallen, module=cfg.module, device=0, init_state="randn", seed=17)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.cuda(0).eval()
outs = []
for n in (1, 33, 512, 1024, 2048, 4096, 5000, 65536 + 777):
    outs.append(m.forward(*synth.feature_batch(n, device="cuda:0", seed=400 + n))[1].cpu().numpy())
np.save(sys.argv[1], np.concatenate(outs))
PY
python /tmp/dump.py /tmp/a.npy && DSP_AMD_LIB=$B python /tmp/dump.py /tmp/b.npy && python -c "
import numpy as np; a=np.load('/tmp/a.npy'); b=np.load('/tmp/b.npy'); print('bit-identical:', a.shape, bool((a.view(np.uint32)==b.view(np.uint32)).all()))"
for b in ${@:-65536}; do for i in 1 2 3; do
  echo "== in-tree, batch $b"; python tools/per_launch.py --batch $b --reps 30 2>/dev/null | grep "lstm_\|sum"
  echo "== variant B, batch $b"; DSP_AMD_LIB=$B python tools/per_launch.py --batch $b --reps 30 2>/dev/null | grep "lstm_\|sum"
done; done
