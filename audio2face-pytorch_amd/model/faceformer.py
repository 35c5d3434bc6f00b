"""Drop-in ``Faceformer`` (reference: /root/reference/src/model/faceformer.py:92-191) on the HIP hot path.

Same constructor signature, sub-module / parameter names and shapes (``audio_encoder.*`` follows HF wav2vec2 keys,
``transformer_decoder.layers.0.*`` follows nn.TransformerDecoderLayer), zero-initialised vertice maps, and
``forward(audio, one_hot, template, **kwargs)`` -> (1, T, V, 3) / ``predict``.

What differs is how the same function is evaluated: the reference re-runs the decoder and the
64->15069 map on the growing prefix every step (O(T^3), faceformer.py:154-185); here the decoder is ONE persistent HIP
kernel per sequence with a K/V cache, the diagonal cross-attention is a per-frame vector computed by two GEMMs, the
feedback ``vertice_map(vertice_map_r(y))`` is the collapsed 64x64 map ``A y + c``, and the 15069-wide head runs once
per frame (fused with the loss in training).  Closed forms replace the materialised (4,600,600) bias table and the
660-row PPE buffer, so T is not limited to 600.

Train mode (what Lightning puts the reference's module in): the encoder applies the reference's dropout / LayerDrop / SpecAugment
(model/wav2vec.py, pinned by fixture G13); the decoder / PPE dropout 0.1 is applied with one mask per frame ROW instead of a fresh
mask per loop step (the reference's redraw makes row j at step i differ from row j at step j - a function with no incremental
evaluation; per-row masks keep the per-step distribution and make the reference's own loop equal this recurrence: fixture G14).
`.eval()` - and `decoder_regularisers = False` / `audio_encoder.train_regularisers = False` - give the deterministic function.

The constructor does not download anything: the reference's ``from_pretrained`` calls need the network; load
pretrained wav2vec2 weights with ``model.audio_encoder.load_state_dict(...)`` (HF key names).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..functional import FFDecoderFn, HeadFn, HeadLossFusedFn, LinearFn, MatmulFn
from .wav2vec import Wav2Vec2Model, normalize_waveform


class PeriodicPositionalEncoding(nn.Module):
    """faceformer.py:70-88; kept as a module so the ``PPE.pe`` buffer exists under the reference's name."""

    def __init__(self, d_model, dropout=0.1, period=25, max_seq_len=600):
        super().__init__()
        pe = torch.zeros(period, d_model)
        position = torch.arange(0, period, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.period = period
        self.register_buffer("pe", pe.unsqueeze(0).repeat(1, max_seq_len // period + 1, 1))

    def table(self):
        return self.pe[0, : self.period]          # one period; the kernel indexes it with i mod period



class Faceformer(nn.Module):
    def __init__(self, n_verts: int, n_onehot: int, audio_encoder: nn.Module | None = None):
        super().__init__()
        self.feature_dim = 64
        self.n_onehot = n_onehot
        self.dataset = "vocaset"
        self.period = 60
        self.vertice_dim = n_verts
        self.audio_encoder = audio_encoder if audio_encoder is not None else Wav2Vec2Model()
        self.audio_feature_map = nn.Linear(768, self.feature_dim)
        self.vertice_map = nn.Linear(self.vertice_dim, self.feature_dim)
        self.PPE = PeriodicPositionalEncoding(self.feature_dim, period=self.period)
        decoder_layer = nn.TransformerDecoderLayer(d_model=self.feature_dim, nhead=4, dim_feedforward=2 * self.feature_dim,
                                                   batch_first=True)
        self.transformer_decoder = nn.TransformerDecoder(decoder_layer, num_layers=1)      # parameter container only
        self.vertice_map_r = nn.Linear(self.feature_dim, self.vertice_dim)
        self.obj_vector = nn.Linear(self.n_onehot, self.feature_dim, bias=False)
        for lin in (self.vertice_map_r, self.vertice_map):                                  # faceformer.py:132-135
            nn.init.constant_(lin.weight, 0)
            nn.init.constant_(lin.bias, 0)

    # ------------------------------------------------------------------ pieces
    def _packed_decoder_params(self):
        """Decoder weights + collapsed feedback in the C-ABI block layout (a2f_hip.h: A2F_FF_*); autograd flows through
        the concatenation back to the individual parameters."""
        L = self.transformer_decoder.layers[0]
        Wm, bm = self.vertice_map.weight, self.vertice_map.bias
        Wr, br = self.vertice_map_r.weight, self.vertice_map_r.bias
        A = MatmulFn.apply(Wm, Wr)                                          # (64,64) = Wm Wr
        c = MatmulFn.apply(Wm, br.reshape(-1, 1)).reshape(-1) + bm           # Wm br + bm
        parts = [L.self_attn.in_proj_weight, L.self_attn.in_proj_bias, L.self_attn.out_proj.weight, L.self_attn.out_proj.bias,
                 L.linear1.weight, L.linear1.bias, L.linear2.weight, L.linear2.bias,
                 L.norm1.weight, L.norm1.bias, L.norm2.weight, L.norm2.bias, L.norm3.weight, L.norm3.bias, A, c]
        return torch.cat([p.reshape(-1) for p in parts])

    # Mixed precision of the reference's trainer (`percision: "16-mixed"`, config.yaml:6 -> Lightning autocast) applies to the wav2vec2
    # encoder only: set to torch.bfloat16 / torch.float16 (registry.build_from_config does it from the config key).  On the hand-written
    # encoder it means plain bf16 operands on the matrix pipe (fp32 accumulate / softmax / LayerNorm) instead of the bf16x3 split
    # products - forward AND backward: every encoder autograd node records the precision it ran under (encoder_ops._save_prec).  The HIP
    # decoder / head / loss always compute in fp32 on the fp32 encoder states.
    encoder_autocast = None

    def encode(self, audio, frame_num):
        wave = normalize_waveform(audio)                                     # audio_preprocessor, faceformer.py:142-144
        if self.encoder_autocast is not None:
            if isinstance(self.audio_encoder, Wav2Vec2Model):
                from .. import ops
                be = ops.backend()
                old, be.mm_precision = be.mm_precision, 0
                try:
                    return self.audio_encoder(wave, self.dataset, frame_num=frame_num).last_hidden_state
                finally:
                    be.mm_precision = old
            with torch.autocast(wave.device.type, dtype=self.encoder_autocast):      # a caller-supplied torch-ops encoder module
                h = self.audio_encoder(wave, self.dataset, frame_num=frame_num).last_hidden_state
            return h.float()
        return self.audio_encoder(wave, self.dataset, frame_num=frame_num).last_hidden_state

    # Train mode: decoder / PPE dropout 0.1 (faceformer.py:73,88,121-127) with one mask per frame row (csrc/ff_decoder.hip explains why
    # not per loop step; fixture G14 pins it), drawn per forward from torch's CPU generator or injected (tests);
    # `decoder_regularisers = False` keeps the deterministic function in train mode.  The encoder's own regularisers: model/wav2vec.py.
    decoder_dropout = 0.1
    decoder_regularisers = True
    injected_decoder_masks = None

    def _decoder_masks(self, S, T, device):
        if not (self.training and self.decoder_regularisers and self.decoder_dropout > 0.0):
            return None
        if self.injected_decoder_masks is not None:
            return self.injected_decoder_masks
        from .. import ops
        from ..dropout import draw_decoder_masks
        static = getattr(self.audio_encoder, "static_rand", None)
        seed = static.dec_seed if static is not None else int(torch.randint(0, 2 ** 62, (1,)).item())      # captured step: the seed lives on the device
        return draw_decoder_masks(ops.backend(), S, T, self.decoder_dropout, seed, device)

    def decode_hidden(self, enc, one_hot):
        """enc (S,T,768), one_hot (S,n_onehot) -> decoder rows y (S,T,64)."""
        S, T, _ = enc.shape
        L = self.transformer_decoder.layers[0]
        d = self.feature_dim
        dm = self.last_decoder_masks = self._decoder_masks(S, T, enc.device)
        mem = LinearFn.apply(enc.reshape(S * T, -1), self.audio_feature_map.weight, self.audio_feature_map.bias)     # :152
        mw, mb = L.multihead_attn.in_proj_weight, L.multihead_attn.in_proj_bias
        v = LinearFn.apply(mem, mw[2 * d:], mb[2 * d:])                      # diagonal enc-dec mask: softmax over one key
        if dm is not None:                                                    # ... whose probability (1) the attention dropout keeps or drops per head
            from ..encoder_ops import DropoutFn
            v = DropoutFn.apply(v, dm.cross_v, dm.scale)
        cross = LinearFn.apply(v, L.multihead_attn.out_proj.weight, L.multihead_attn.out_proj.bias)
        if dm is not None:
            cross = DropoutFn.apply(cross, dm.cross_out, dm.scale)            # dropout2
        style = LinearFn.apply(one_hot, self.obj_vector.weight, None)       # (S,64)  :148
        return FFDecoderFn.apply(self._packed_decoder_params(), self.PPE.table(), cross.view(S, T, d), style, self.period,
                                 (dm.row, dm.attn, dm.scale) if dm is not None else None)

    last_decoder_masks = None

    # ------------------------------------------------------------------ reference surface
    def forward(self, audio, one_hot, template, **kwargs):
        frame_num = audio.shape[1] * 60 // 16000                             # faceformer.py:141
        enc = kwargs.get("encoder_states")
        if enc is None:
            enc = self.encode(audio, frame_num)
        y = self.decode_hidden(enc, one_hot)                                 # (1,T,64)
        T = y.shape[1]
        pred = HeadFn.apply(y.reshape(T, -1), self.vertice_map_r.weight, self.vertice_map_r.bias, template.reshape(1, -1))
        return pred.view(1, T, -1, 3)

    def predict(self, audio, one_hot, template, **kwargs):
        return self(audio, one_hot, template, **kwargs)

    def forward_with_loss(self, audio, one_hot, template, gt, k_rec: float = 1.0, k_vel: float = 10.0, unit_grad: bool = False, **kwargs):
        """Training fast path = forward + FaceFormerLoss (loss.py:4-17) with the head, the loss and the head backward
        fused into one pass; the odd tail frame is predicted but left out of the loss, as the reference does."""
        frame_num = audio.shape[1] * 60 // 16000
        enc = kwargs.get("encoder_states")
        if enc is None:
            enc = self.encode(audio, frame_num)
        y = self.decode_hidden(enc, one_hot)
        T = y.shape[1]
        if T < 2:
            raise RuntimeError("FaceFormerLoss needs at least 2 frames")
        pred, out3 = HeadLossFusedFn.apply(y.reshape(T, -1), self.vertice_map_r.weight, self.vertice_map_r.bias,
                                           template.reshape(1, -1), gt.reshape(T, -1), T - (T % 2), k_rec, k_vel, unit_grad)
        return pred.view(1, T, -1, 3), {"loss": out3[0], "rec_loss": out3[1].detach(), "vel_loss": out3[2].detach()}

    def _head_loss_batch(self, y, template, gt, k_rec, k_vel):
        """y (S,T,64) -> (pred (S,T,V,3), out3 = [loss, rec, vel] averaged over the S sequences).  Even T: all S * T frames in ONE
        fused-head launch with one template row per sequence (the velocity pairs (2k, 2k+1) never straddle two sequences and the
        mean over S * T frames equals the mean of the per-sequence means); odd T: one launch per sequence, each dropping its own
        tail frame as loss.py:13-15 does."""
        S, T = y.shape[0], y.shape[1]
        if T % 2 == 0:
            p, tot = HeadLossFusedFn.apply(y.reshape(S * T, -1), self.vertice_map_r.weight, self.vertice_map_r.bias, template.reshape(S, -1),
                                           gt.reshape(S * T, -1), S * T, k_rec, k_vel)
            return p.view(S, T, -1, 3), tot
        preds, tot = [], None
        for s in range(S):
            p, o3 = HeadLossFusedFn.apply(y[s], self.vertice_map_r.weight, self.vertice_map_r.bias, template[s].reshape(1, -1),
                                          gt[s].reshape(T, -1), T - (T % 2), k_rec, k_vel)
            preds.append(p)
            tot = o3 if tot is None else tot + o3
        return torch.stack(preds).view(S, T, -1, 3), tot / S

    def forward_with_loss_batch(self, audio, one_hot, template, gt, k_rec: float = 1.0, k_vel: float = 10.0):
        """S independent batch-1 problems in one step (the reference trains batch = 1, train.py:33):
        audio (S,N), one_hot (S,n_onehot), template (S,V,3), gt (S,T,V,3).  The encoder runs batched, the decoder
        recurrence launches one workgroup per sequence, the fused head takes all frames at once (`_head_loss_batch`).
        Returns (pred (S,T,V,3), mean loss dict)."""
        frame_num = audio.shape[1] * 60 // 16000
        enc = self.encode(audio, frame_num)
        y = self.decode_hidden(enc, one_hot)                                  # (S,T,64)
        pred, tot = self._head_loss_batch(y, template, gt, k_rec, k_vel)
        return pred, {"loss": tot[0], "rec_loss": tot[1].detach(), "vel_loss": tot[2].detach()}

    _pipe_stream = None
    pipeline_parts = 2            # parts of the pipelined step (the measurements in train_step_pipelined; tests exercise 4)

    @staticmethod
    def grad_bucket_key(name: str):
        """Data-parallel gradient buckets in parameter order (trainer.BucketedAllReduce): the encoder's front end (feature extractor,
        feature projection, positional convolution), one bucket per Transformer layer, the decoder side - the order in which the backward
        finishes them, reversed."""
        if name.startswith("audio_encoder.encoder.layers."):
            return "layer" + name.split(".")[3]
        return "front" if name.startswith("audio_encoder.") else "decoder"

    def train_step_pipelined(self, audio, one_hot, template, gt, k_rec: float = 1.0, k_vel: float = 10.0, before_last_backward=None):
        """forward + FaceFormerLoss + backward of S >= 2 sequences (gradients ACCUMULATE into .grad, as loss.backward() does) with
        the latency-bound decoder recurrence hidden under encoder work.  The recurrence occupies one workgroup per sequence for
        ~17 ms (forward + backward at T = 600) whatever S is; run back to back with the encoder it leaves >= 75 % of the CUs idle
        for that time.  Here the step is cut into two half-batches A, B and issued on two HIP streams:

            main stream   : encoder fwd A | encoder fwd B |     encoder bwd A      | encoder bwd B
            side stream A :               | decoder fwd + head + decoder bwd of A  |
            side stream B :                               | decoder fwd + head of B, then (after A's) its backward |

        so A's recurrence runs beside B's encoder forward and B's beside A's encoder backward.  To issue a half's decoder backward
        BEFORE the other half's encoder launches (the host issues ~2 500 launches per step; what is issued late starts late) the
        autograd graph of a half is cut at the encoder output: the decoder side runs on a detached copy and is differentiated
        right after its forward (`backward(inputs=...)` on the side stream); d loss / d encoder-output then seeds the encoder
        backward on the main stream.  The arithmetic per sequence is that of `forward_with_loss_batch` (same kernels); only the order
        in which the two halves' gradients are added differs.  Returns the loss dict (detached values, mean over the S sequences)."""
        S = audio.shape[0]
        assert S >= 2 and audio.is_cuda
        frame_num = audio.shape[1] * 60 // 16000
        cur = torch.cuda.current_stream()
        if Faceformer._pipe_stream is None or Faceformer._pipe_stream[0].device != audio.device:
            # one side stream per half: the recurrence takes the same time whatever the number of sequences, so at long T (3000 frames: 89 ms) the
            # second half's recurrence has to start while the first half's is still running, not queue behind it
            Faceformer._pipe_stream = [torch.cuda.Stream(device=audio.device, priority=-1) for _ in range(2)]     # short dependent chains go first in dispatch
        sides = Faceformer._pipe_stream
        if frame_num < 1200:
            # at 600 frames a half's recurrence (11 ms) is over long before the other half's starts: one side stream does - and a further stream
            # can land on a hardware queue another stream of the process already uses (measured inside the full bench run: 182 vs 157 ms)
            sides = [sides[0], sides[0]]
        # Parts: two equal halves (measured best at 600 frames: 7/16 .. 5/16 for the first half +0.4 .. +1.4 %).  More parts would start the first
        # recurrence earlier (at long T a part's recurrence outlasts a half's encoder forward), but measured at 12 x 3000 they lose more in the
        # products than they win: 2 parts 239.1 ms, 3 parts 248.4, 4 parts 250.5, 6 parts 299.0 (same box) - A2F_FF_PIPELINE_PARTS keeps the hook.
        parts = max(2, min(int(self.pipeline_parts), S))
        cuts = [(S * g + parts - 1) // parts for g in range(parts)] + [S]
        enc_params = {id(p) for p in self.audio_encoder.parameters()}
        dec_params = [p for p in self.parameters() if p.requires_grad and id(p) not in enc_params]
        halves, outs = [], []
        for g in range(parts):
            lo, hi = cuts[g], cuts[g + 1]
            enc = self.encode(audio[lo:hi], frame_num)                          # main stream
            ready = cur.record_event()
            side = sides[g % 2]
            with torch.cuda.stream(side):
                side.wait_event(ready)
                enc.record_stream(side)
                enc_d = enc.detach().requires_grad_(True)
                y = self.decode_hidden(enc_d, one_hot[lo:hi])
                _pred, out3 = self._head_loss_batch(y, template[lo:hi], gt[lo:hi], k_rec, k_vel)
                w = (hi - lo) / S
                if halves:
                    side.wait_event(halves[-1][2])                              # the halves add into the same decoder-side .grad tensors: in half order
                torch.autograd.backward(out3[0] * w, inputs=[enc_d] + dec_params)
                outs.append(out3.detach() * w)
                halves.append((enc, enc_d.grad, side.record_event()))
                del y, _pred, out3, enc_d
        # (Rounds 4-5: with the last part's decoder forward + head beside the encoder backward, two runs of the step differed in their last bits once in ~10.  It
        # was no missing dependency: one packed-fp32 instruction form in layernorm_bwd_kernel misread a register while other queues' kernels overlapped it -
        # DESIGN 2.4; the library is built without that form (_build.FLAGS), the full overlap is the only order again.)
        for i, (enc, d_enc, done) in enumerate(halves):
            cur.wait_event(done)
            d_enc.record_stream(cur)
            if before_last_backward is not None and i == len(halves) - 1:
                before_last_backward()                                          # the gradient exchange may start: every earlier pass has been accumulated
            enc.backward(d_enc)                                                 # main stream: encoder backward of this half
        halves.clear()
        for side in sides[:1] if sides[0] is sides[1] else sides:
            cur.wait_stream(side)
        for o in outs:
            o.record_stream(cur)
        tot = outs[0]
        for o in outs[1:]:
            tot = tot + o
        return {"loss": tot[0], "rec_loss": tot[1], "vel_loss": tot[2]}
