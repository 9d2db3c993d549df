"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The shipped path (deepsignal_plant_amd) never imports anything from oracle/.

What is restated, with the reference lines it follows (paths relative to /root/reference):

* ``ModelBiLSTM.forward``                      deepsignal_plant/models.py:178-240
* ``ModelBiLSTM.__init__`` parameter shapes     deepsignal_plant/models.py:103-164
* ``ModelBiLSTM.init_hidden`` (shape/order)     deepsignal_plant/models.py:169-176
* ``torch.nn.LSTM`` cell semantics (third-party dependency, torch pinned by the reference at
  ``torch>=1.2.0,<=1.11.0`` in requirements.txt:5; the algorithm is the published PyTorch LSTM
  definition): gates packed i,f,g,o along 4H; ``gates = x W_ih^T + b_ih + h W_hh^T + b_hh``;
  ``c' = sigmoid(f) c + sigmoid(i) tanh(g)``; ``h' = sigmoid(o) tanh(c')``; the ``_reverse``
  direction runs t = L-1 .. 0 and writes out[:, t]; layer output = [fwd | bwd]; h0[2*layer+dir].

Parity pinning: the reference holds no tests or golden vectors for this path,
so this oracle is pinned by fixtures generated in the build container by importing the reference
itself (tests/golden/make_golden.py -> tests/golden/*.npz), with the LSTM initial states pinned
because the reference draws them from torch.randn on every call (models.py:169-176).
"""
from __future__ import annotations

import numpy as np

MODULES = ("both_bilstm", "seq_bilstm", "signal_bilstm")


class OracleConfig:
    """Mirror of the constructor arguments at deepsignal_plant/models.py:103-106."""

    def __init__(self, seq_len=13, signal_len=16, num_layers1=3, num_layers2=1, num_classes=2,
                 hidden_size=256, vocab_size=16, embedding_size=4, is_base=True, is_signallen=True,
                 module="both_bilstm"):
        if module not in MODULES:
            raise ValueError("--model_type is not right!")  # models.py:127-128
        self.seq_len = int(seq_len)
        self.signal_len = int(signal_len)
        self.num_layers1 = int(num_layers1)
        self.num_layers2 = int(num_layers2)
        self.num_classes = int(num_classes)
        self.hidden_size = int(hidden_size)
        self.vocab_size = int(vocab_size)
        self.embedding_size = int(embedding_size)
        self.is_base = bool(is_base)
        self.is_signallen = bool(is_signallen)
        self.module = module
        # models.py:120-126
        self.nhid_seq = 0
        self.nhid_signal = 0
        if module == "both_bilstm":
            self.nhid_seq = self.hidden_size // 2
            self.nhid_signal = self.hidden_size - self.nhid_seq
        elif module == "seq_bilstm":
            self.nhid_seq = self.hidden_size
        else:
            self.nhid_signal = self.hidden_size
        self.sigfea_num = 3 if self.is_signallen else 2  # models.py:135
        self.seq_in = (self.embedding_size if self.is_base else 0) + self.sigfea_num  # :136-141

    def as_dict(self):
        return dict(seq_len=self.seq_len, signal_len=self.signal_len, num_layers1=self.num_layers1,
                    num_layers2=self.num_layers2, num_classes=self.num_classes,
                    hidden_size=self.hidden_size, vocab_size=self.vocab_size,
                    embedding_size=self.embedding_size, is_base=self.is_base,
                    is_signallen=self.is_signallen, module=self.module)


def _lstm_spec(prefix, in_size, hid, layers):
    out = []
    for k in range(layers):
        isz = in_size if k == 0 else 2 * hid
        for suf in ("", "_reverse"):
            out.append(("%s.weight_ih_l%d%s" % (prefix, k, suf), (4 * hid, isz)))
            out.append(("%s.weight_hh_l%d%s" % (prefix, k, suf), (4 * hid, hid)))
            out.append(("%s.bias_ih_l%d%s" % (prefix, k, suf), (4 * hid,)))
            out.append(("%s.bias_hh_l%d%s" % (prefix, k, suf), (4 * hid,)))
    return out


def state_dict_spec(cfg: OracleConfig):
    """(name, shape) in the order torch's state_dict() lists them for models.py:130-161."""
    spec = []
    if cfg.module != "signal_bilstm":
        spec.append(("embed.weight", (cfg.vocab_size, cfg.embedding_size)))
        spec += _lstm_spec("lstm_seq", cfg.seq_in, cfg.nhid_seq, cfg.num_layers2)
        spec += [("fc_seq.weight", (cfg.nhid_seq, 2 * cfg.nhid_seq)), ("fc_seq.bias", (cfg.nhid_seq,))]
    if cfg.module != "seq_bilstm":
        spec += _lstm_spec("lstm_signal", cfg.signal_len, cfg.nhid_signal, cfg.num_layers2)
        spec += [("fc_signal.weight", (cfg.nhid_signal, 2 * cfg.nhid_signal)),
                 ("fc_signal.bias", (cfg.nhid_signal,))]
    spec += _lstm_spec("lstm_comb", cfg.hidden_size, cfg.hidden_size, cfg.num_layers1)
    spec += [("fc1.weight", (cfg.hidden_size, 2 * cfg.hidden_size)), ("fc1.bias", (cfg.hidden_size,)),
             ("fc2.weight", (cfg.num_classes, cfg.hidden_size)), ("fc2.bias", (cfg.num_classes,))]
    return spec


def make_weights(cfg: OracleConfig, seed: int, scale: float = 1.0):
    """Build-defined deterministic weights: per tensor, in state_dict order,
    U(-k, k)*scale with k = 1/sqrt(H) for LSTM tensors, 1/sqrt(fan_in) for Linear, N(0,1) for the
    embedding (the PyTorch default-init scales).  Only the seed is committed with a fixture."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in state_dict_spec(cfg):
        if name == "embed.weight":
            a = rng.standard_normal(shape)
        elif name.startswith("lstm"):
            hid = shape[0] // 4
            k = 1.0 / np.sqrt(hid)
            a = rng.uniform(-k, k, size=shape)
        else:
            mod = name.split(".")[0]
            fan_in = {"fc_seq": 2 * cfg.nhid_seq, "fc_signal": 2 * cfg.nhid_signal,
                      "fc1": 2 * cfg.hidden_size, "fc2": cfg.hidden_size}[mod]
            k = 1.0 / np.sqrt(fan_in)
            a = rng.uniform(-k, k, size=shape)
        w[name] = np.ascontiguousarray((a * scale).astype(np.float32))
    return w


def make_inputs(cfg: OracleConfig, n: int, seed: int, wide_alphabet: bool = False):
    """Synthetic feature rows as SURVEY.md 8(d) specifies (the extract format of
    deepsignal_plant/extract_features.py:232-251, :381-395): kmer codes uniform{0..3} with the centre
    forced to C=1; means N(0,1) 6dp; stds |N(.25,.1)| 6dp; lens int U[2,40); signals N(0,1) 6dp with
    centred zero padding where len < signal_len (left = pad//2)."""
    rng = np.random.default_rng(seed)
    L, S = cfg.seq_len, cfg.signal_len
    hi = cfg.vocab_size if wide_alphabet else 4
    kmer = rng.integers(0, hi, size=(n, L)).astype(np.float32)
    kmer[:, L // 2] = 1.0
    means = np.around(rng.standard_normal((n, L)), 6).astype(np.float32)
    stds = np.around(np.abs(rng.normal(0.25, 0.1, size=(n, L))), 6).astype(np.float32)
    lens_i = rng.integers(2, 40, size=(n, L))
    sig = np.around(rng.standard_normal((n, L, S)), 6)
    idx = np.arange(S)[None, None, :]
    ln = np.minimum(lens_i, S)[:, :, None]
    left = (S - ln) // 2
    mask = (idx >= left) & (idx < left + ln)
    signals = np.where(mask, sig, 0.0).astype(np.float32)
    return kmer, means, stds, lens_i.astype(np.float32), signals


def make_extreme_inputs(cfg: OracleConfig, n: int, seed: int):
    """Legal but extreme feature rows (what `extract` can emit, extract_features.py:232-251, :381-395): the whole IUPAC
    alphabet; bases of 1 sample and stalls of 130..420 samples; per-base means / signals that are ordinary, tiny, or
    huge (|x| 1e3..1e6, e.g. an un-normalised read) of both signs; stds from 0 to 1e3; signal rectangles that are all
    padding but one sample.  Saturates the front-end LSTM gates (exp2 overflow / underflow of the activations)."""
    rng = np.random.default_rng(seed)
    L, S = cfg.seq_len, cfg.signal_len
    kmer = rng.integers(0, cfg.vocab_size, size=(n, L)).astype(np.float32)
    kmer[:, L // 2] = 1.0
    kind = rng.integers(0, 4, size=(n, L))                       # 0 ordinary, 1 tiny, 2 huge, 3 huge negative
    scale = np.choose(kind, [1.0, 1e-6, 1.0, 1.0])
    shift = np.choose(kind, [0.0, 0.0, 1.0, -1.0]) * 10.0 ** rng.uniform(3, 6, size=(n, L))
    means = np.around(rng.standard_normal((n, L)) * scale + shift, 6).astype(np.float32)
    stds = np.around(np.abs(rng.normal(0.25, 0.1, size=(n, L))) * np.choose(kind, [1.0, 0.0, 1e3, 40.0]), 6).astype(np.float32)
    lens_i = np.where(rng.random((n, L)) < 0.15, rng.integers(130, 421, size=(n, L)),
                      np.where(rng.random((n, L)) < 0.2, 1, rng.integers(2, 40, size=(n, L))))
    sig = np.around(rng.standard_normal((n, L, S)) * scale[:, :, None] + shift[:, :, None], 6)
    idx = np.arange(S)[None, None, :]
    ln = np.minimum(lens_i, S)[:, :, None]
    left = (S - ln) // 2
    signals = np.where((idx >= left) & (idx < left + ln), sig, 0.0).astype(np.float32)
    return kmer, means, stds, lens_i.astype(np.float32), signals


def init_state_shapes(cfg: OracleConfig, n: int):
    """Shapes and draw order of init_hidden: models.py:169-176, called at :196-198, :212-214, :226-228."""
    shp = []
    if cfg.module != "signal_bilstm":
        shp += [("h_seq", (2 * cfg.num_layers2, n, cfg.nhid_seq)), ("c_seq", (2 * cfg.num_layers2, n, cfg.nhid_seq))]
    if cfg.module != "seq_bilstm":
        shp += [("h_sig", (2 * cfg.num_layers2, n, cfg.nhid_signal)),
                ("c_sig", (2 * cfg.num_layers2, n, cfg.nhid_signal))]
    shp += [("h_comb", (2 * cfg.num_layers1, n, cfg.hidden_size)), ("c_comb", (2 * cfg.num_layers1, n, cfg.hidden_size))]
    return shp


def make_init_states(cfg: OracleConfig, n: int, seed: int):
    """Pinned N(0,1) initial states from a build-defined generator, in init_hidden's draw order."""
    rng = np.random.default_rng(seed)
    return {k: rng.standard_normal(s).astype(np.float32) for k, s in init_state_shapes(cfg, n)}


def zero_init_states(cfg: OracleConfig, n: int):
    return {k: np.zeros(s, np.float32) for k, s in init_state_shapes(cfg, n)}


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_bidir(x, w, prefix, layers, hid, h0, c0, dtype):
    """x [N,L,I] -> [N,L,2*hid]; torch.nn.LSTM(batch_first, bidirectional) semantics (models.py:137-157)."""
    n, L, _ = x.shape
    inp = x
    for k in range(layers):
        out = np.zeros((n, L, 2 * hid), dtype)
        for d, suf in enumerate(("", "_reverse")):
            wih = w["%s.weight_ih_l%d%s" % (prefix, k, suf)].astype(dtype)
            whh = w["%s.weight_hh_l%d%s" % (prefix, k, suf)].astype(dtype)
            b = (w["%s.bias_ih_l%d%s" % (prefix, k, suf)].astype(dtype) +
                 w["%s.bias_hh_l%d%s" % (prefix, k, suf)].astype(dtype))
            h = h0[2 * k + d].astype(dtype)
            c = c0[2 * k + d].astype(dtype)
            steps = range(L) if d == 0 else range(L - 1, -1, -1)
            for t in steps:
                g = inp[:, t, :] @ wih.T + h @ whh.T + b
                i_, f_, g_, o_ = g[:, :hid], g[:, hid:2 * hid], g[:, 2 * hid:3 * hid], g[:, 3 * hid:]
                c = _sigmoid(f_) * c + _sigmoid(i_) * np.tanh(g_)
                h = _sigmoid(o_) * np.tanh(c)
                out[:, t, d * hid:(d + 1) * hid] = h
        inp = out
    return inp


def forward(cfg: OracleConfig, w, kmer, means, stds, lens, signals, states, dtype=np.float64,
            want_intermediates=False):
    """Restates ModelBiLSTM.forward (models.py:178-240). Returns (logits, probs[, intermediates])."""
    n = kmer.shape[0]
    L = cfg.seq_len
    inter = {}
    out_seq = out_sig = None
    if cfg.module != "signal_bilstm":
        m = means.reshape(n, L, 1).astype(dtype)  # :182-184
        s = stds.reshape(n, L, 1).astype(dtype)
        ln = lens.reshape(n, L, 1).astype(dtype)
        feats = []
        if cfg.is_base:
            feats.append(w["embed.weight"].astype(dtype)[kmer.astype(np.int64)])  # :186
        feats += [m, s]
        if cfg.is_signallen:
            feats.append(ln)
        x = np.concatenate(feats, axis=2)  # :187-195
        o = lstm_bidir(x, w, "lstm_seq", cfg.num_layers2, cfg.nhid_seq, states["h_seq"], states["c_seq"], dtype)
        inter["lstm_seq"] = o
        o = o @ w["fc_seq.weight"].astype(dtype).T + w["fc_seq.bias"].astype(dtype)  # :199
        out_seq = np.maximum(o, 0)  # :201
        inter["out_seq"] = out_seq
    if cfg.module != "seq_bilstm":
        x = signals.astype(dtype)  # :206
        o = lstm_bidir(x, w, "lstm_signal", cfg.num_layers2, cfg.nhid_signal, states["h_sig"], states["c_sig"], dtype)
        inter["lstm_signal"] = o
        o = o @ w["fc_signal.weight"].astype(dtype).T + w["fc_signal.bias"].astype(dtype)  # :215
        out_sig = np.maximum(o, 0)  # :217
        inter["out_signal"] = out_sig
    if cfg.module == "seq_bilstm":  # :220-225
        out = out_seq
    elif cfg.module == "signal_bilstm":
        out = out_sig
    else:
        out = np.concatenate((out_seq, out_sig), axis=2)
    out = lstm_bidir(out, w, "lstm_comb", cfg.num_layers1, cfg.hidden_size, states["h_comb"], states["c_comb"], dtype)
    inter["lstm_comb"] = out
    H = cfg.hidden_size
    feat = np.concatenate((out[:, -1, :H], out[:, 0, H:]), axis=1)  # :229-231
    o = feat @ w["fc1.weight"].astype(dtype).T + w["fc1.bias"].astype(dtype)  # :235 (dropouts are identity in eval)
    o = np.maximum(o, 0)  # :237
    logits = o @ w["fc2.weight"].astype(dtype).T + w["fc2.bias"].astype(dtype)  # :238
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    probs = e / e.sum(axis=1, keepdims=True)  # :240 Softmax(1)
    if want_intermediates:
        return logits, probs, inter
    return logits, probs


def flops_per_site(cfg: OracleConfig):
    """MACs per site: per BiLSTM layer 2*L*4H*(I+H); Linear L*out*in; x2 for FLOP."""
    L = cfg.seq_len
    mac = 0

    def lstm(i, h, layers):
        m = 0
        for k in range(layers):
            isz = i if k == 0 else 2 * h
            m += 2 * L * 4 * h * (isz + h)
        return m
    if cfg.module != "signal_bilstm":
        mac += lstm(cfg.seq_in, cfg.nhid_seq, cfg.num_layers2) + L * cfg.nhid_seq * 2 * cfg.nhid_seq
    if cfg.module != "seq_bilstm":
        mac += lstm(cfg.signal_len, cfg.nhid_signal, cfg.num_layers2) + L * cfg.nhid_signal * 2 * cfg.nhid_signal
    mac += lstm(cfg.hidden_size, cfg.hidden_size, cfg.num_layers1)
    mac += cfg.hidden_size * 2 * cfg.hidden_size + cfg.num_classes * cfg.hidden_size
    return 2 * mac
