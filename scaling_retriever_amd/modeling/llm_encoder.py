"""Host-side mirror of the reference's encoder API over the HIP engine.

Same names, arguments and error behaviour as
/root/reference/scaling_retriever/modeling/llm_encoder.py (LLM2Retriever :14-150,
DecoderOnlyBiSparse :175-196, DecoderOnlyBiDense :370-520, LlamaBiSparse :199-201,
LlamaBiDense :523-525) for the inference path: load / load_from_lora / encode /
query_encode / doc_encode, attributes vocab_size / hidden_size / T / base_model.config.

All arithmetic runs in libsr_hip.so (csrc/encoder.hip, gemm_bf16.hip, attention.hip).
torch is used for device memory and checkpoint I/O only; there is no eager fallback -
without the HIP library or a ROCm device, encode() raises.
Training-only methods of the reference (forward losses, build with LoRA adapters,
gradient checkpointing) are out of scope.
"""
import ctypes
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from .. import _lib

TARGET_MODULES = ["q_proj", "v_proj", "o_proj", "k_proj", "down_proj", "up_proj", "gate_proj"]


# --------------------------------------------------------------------------- config
class LlamaConfigLite(SimpleNamespace):
    """The fields of an HF LlamaConfig the hot path reads (attribute access like the HF object)."""

    @classmethod
    def from_dict(cls, d):
        d = dict(d)
        nh = d["num_attention_heads"]
        d.setdefault("num_key_value_heads", nh)
        if d.get("num_key_value_heads") is None:
            d["num_key_value_heads"] = nh
        if not d.get("head_dim"):
            d["head_dim"] = d["hidden_size"] // nh
        d.setdefault("rms_norm_eps", 1e-6)
        d.setdefault("rope_theta", 10000.0)
        d.setdefault("tie_word_embeddings", False)
        d.setdefault("rope_scaling", None)
        return cls(**d)

    def to_dict(self):
        return dict(self.__dict__)


def _canonical_name(k):
    """Checkpoint key -> 'model.*' / 'lm_head.weight' naming (bare LlamaModel checkpoints lack the prefix)."""
    if k.startswith("lm_head."):
        return k
    return k if k.startswith("model.") else "model." + k


def _read_checkpoint(path):
    """name -> torch tensor (CPU) for every tensor in an HF checkpoint directory."""
    from safetensors import safe_open
    idx = os.path.join(path, "model.safetensors.index.json")
    files = []
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    elif os.path.exists(os.path.join(path, "model.safetensors")):
        files = ["model.safetensors"]
    out = {}
    if files:
        for fn in files:
            with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
                for k in f.keys():
                    out[_canonical_name(k)] = f.get_tensor(k)
        return out
    binf = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(binf):
        sd = torch.load(binf, map_location="cpu", weights_only=True)
        return {_canonical_name(k): v for k, v in sd.items()}
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")


def _resolve_dir(name_or_path, access_token=None):
    if os.path.isdir(name_or_path):
        return name_or_path
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(name_or_path, token=access_token)
    except Exception as e:  # offline box
        raise FileNotFoundError(f"'{name_or_path}' is not a local directory and could not be fetched from the hub: {e}")


def _load_adapter_state(lora_name_or_path):
    """adapter_model.safetensors / adapter_model.bin, as llm_encoder.py:486-493 reads them."""
    from safetensors.torch import load_file
    if os.path.isdir(lora_name_or_path):
        st = os.path.join(lora_name_or_path, "adapter_model.safetensors")
        if os.path.exists(st):
            return load_file(st)
        bn = os.path.join(lora_name_or_path, "adapter_model.bin")
        if os.path.exists(bn):
            return torch.load(bn, map_location="cpu", weights_only=True)
        raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin under {lora_name_or_path}")
    from huggingface_hub import hf_hub_download
    return torch.load(hf_hub_download(lora_name_or_path, "adapter_model.bin"), map_location="cpu", weights_only=True)


def _adapter_target(key):
    """'base_model.model[.model].layers.N.<mod>.lora_A[.default].weight' -> ('model.layers.N.<mod>.weight', 'A')."""
    k = key
    if k.startswith("base_model.model."):
        k = k[len("base_model.model."):]
    for tag, ab in ((".lora_A.", "A"), (".lora_B.", "B")):
        if tag in k:
            mod = k.split(tag)[0]
            return _canonical_name(mod + ".weight"), ab
    return None, None


# --------------------------------------------------------------------------- backbone
class HipLlamaBackbone(torch.nn.Module):
    """Stands where the reference keeps `base_model` (LlamaBiModel / LlamaBiForMNTP): owns the
    checkpoint tensors until the model is moved to a ROCm device, then the sr_model handle."""

    def __init__(self, config, weights, has_lm_head, lora=None, max_batch_tokens=32768, max_batch_seqs=1024,
                 precision="auto", fp32_planes=None):
        super().__init__()
        # Precision regime per call, as in the reference, where the CALLER decides it with torch.autocast:
        #   "auto" (default)  bf16 regime inside torch.autocast("cuda", dtype=torch.bfloat16) - documents and sparse
        #                     queries (indexer.py:46-52, :255-256, :390-391) - fp32 regime otherwise - dense queries
        #                     (eval_dense.py:94-106) and examples/quick_start.py;
        #   "bf16" / "fp32"   force one regime.
        # fp32_planes: how the fp32 regime represents an fp32 operand: 16 (default) = two fp16 planes of power-of-two scaled
        # rows, 3 plane products per GEMM - the error of an fp32 GEMM; 3 = three bf16 planes, 6 products; 2 = two bf16
        # planes, 3 products (~2^-17); 0 = no fp32 regime and no extra weight copies.  Default 16, or SR_FP32_PLANES.
        if precision not in ("auto", "bf16", "fp32"):
            raise ValueError(f"precision must be 'auto', 'bf16' or 'fp32', got {precision!r}")
        self.precision = precision
        if fp32_planes is None:
            fp32_planes = int(os.environ.get("SR_FP32_PLANES", "16"))
        if fp32_planes not in (0, 2, 3, 16):
            raise ValueError(f"fp32_planes must be 0, 2, 3 or 16, got {fp32_planes}")
        self.fp32_planes = int(fp32_planes)
        self.config = config if isinstance(config, LlamaConfigLite) else LlamaConfigLite.from_dict(config)
        self._weights = weights            # name -> tensor (host or device), dropped after upload
        self._lora = lora                  # {"scale": float, "A": {name: t}, "B": {name: t}} or None
        self.has_lm_head = bool(has_lm_head)
        self.max_batch_tokens = int(max_batch_tokens)
        self.max_batch_seqs = int(max_batch_seqs)
        self._h = None
        self._device = None
        self._lib = None

    # ---- engine -----------------------------------------------------------------
    def _c_config(self):
        c = self.config
        rs = getattr(c, "rope_scaling", None) or getattr(c, "rope_parameters", None)
        llama3 = 0
        fac = lo = hi = 1.0
        old = 0
        if rs:
            rtype = rs.get("rope_type", rs.get("type"))
            if rtype == "llama3":
                llama3 = 1
                fac, lo, hi = float(rs["factor"]), float(rs["low_freq_factor"]), float(rs["high_freq_factor"])
                old = int(rs["original_max_position_embeddings"])
            elif rtype not in (None, "default"):
                raise NotImplementedError(f"rope scaling '{rtype}' is not supported")
        theta = float(rs["rope_theta"]) if (rs and "rope_theta" in rs) else float(c.rope_theta)
        return _lib.SrModelConfig(
            vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
            num_layers=c.num_hidden_layers, num_heads=c.num_attention_heads, num_kv_heads=c.num_key_value_heads,
            head_dim=c.head_dim, rms_norm_eps=float(c.rms_norm_eps), rope_theta=theta, rope_llama3=llama3,
            rope_factor=fac, rope_low_freq_factor=lo, rope_high_freq_factor=hi, rope_original_max_pos=old,
            tie_word_embeddings=int(bool(c.tie_word_embeddings)), has_lm_head=int(self.has_lm_head),
            max_batch_tokens=self.max_batch_tokens, max_batch_seqs=self.max_batch_seqs, fp32_planes=self.fp32_planes)

    def build_engine(self, device):
        _lib.require_gpu()
        device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if device.type != "cuda":
            raise _lib.SrHipError("the encoder runs on ROCm devices only (there is no CPU fallback)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._h is not None:
            if device == self._device:
                return self
            raise _lib.SrHipError("model already lives on %s; moving it to %s is not supported" % (self._device, device))
        if self._weights is None:
            raise _lib.SrHipError("checkpoint tensors were released; reload the model")
        lib = _lib.load()
        with torch.cuda.device(device):
            cfg = self._c_config()
            h = ctypes.c_void_p()
            _lib.check(lib.sr_model_create(ctypes.byref(h), ctypes.byref(cfg)), "sr_model_create")
            try:
                stream = _lib.stream_ptr()
                for name, t in self._weights.items():
                    if name == "lm_head.weight" and not self.has_lm_head:
                        continue
                    if "rotary_emb" in name:
                        continue
                    if isinstance(t, np.ndarray):
                        t = torch.from_numpy(t)
                    lora_hit = self._lora is not None and name in self._lora["A"]
                    if lora_hit:
                        t = t.to(device=device, dtype=torch.float32).contiguous()
                        A = self._lora["A"][name].to(device=device, dtype=torch.float32).contiguous()
                        Bm = self._lora["B"][name].to(device=device, dtype=torch.float32).contiguous()
                        r = A.shape[0]
                        if A.shape[1] != t.shape[1] or Bm.shape != (t.shape[0], r):
                            raise ValueError(f"LoRA shapes do not match {name}: W{tuple(t.shape)} A{tuple(A.shape)} B{tuple(Bm.shape)}")
                        _lib.check(lib.sr_lora_merge(t.data_ptr(), A.data_ptr(), Bm.data_ptr(), t.shape[0], t.shape[1], r,
                                                     float(self._lora["scale"]), stream), "sr_lora_merge")
                    elif t.dtype == torch.bfloat16:
                        t = t.to(device=device).contiguous()
                    else:
                        t = t.to(device=device, dtype=torch.float32).contiguous()
                    dt = _lib.SR_DTYPE_BF16 if t.dtype == torch.bfloat16 else _lib.SR_DTYPE_F32
                    rows, cols = (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.shape[0], 1)
                    _lib.check(lib.sr_model_set_weight(h, name.encode(), t.data_ptr(), dt, rows, cols, stream),
                               f"sr_model_set_weight({name})")
                    torch.cuda.current_stream().synchronize()
                    del t
                _lib.check(lib.sr_model_finalize(h), "sr_model_finalize")
            except Exception:
                lib.sr_model_destroy(h)
                raise
        self._h, self._device, self._lib = h, device, lib
        self._weights = None
        self._lora = None
        return self

    @property
    def device(self):
        return self._device if self._device is not None else torch.device("cpu")

    def to(self, *args, **kwargs):
        device = kwargs.get("device", args[0] if args else None)
        if isinstance(device, (str, int, torch.device)):
            d = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
            if d.type == "cuda":
                self.build_engine(d)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def _ensure_engine(self, like):
        if self._h is None:
            if like is not None and like.is_cuda:
                self.build_engine(like.device)
            elif torch.cuda.is_available():
                self.build_engine(torch.device("cuda", torch.cuda.current_device()))
            else:
                raise _lib.SrHipError("no ROCm device: LlamaBi* encode runs on MI355X only (no CPU fallback)")

    def resolve_precision(self):
        """'bf16' or 'fp32' for a call made right now (see __init__)."""
        if self.precision != "auto":
            return self.precision
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            if dt != torch.bfloat16:
                raise NotImplementedError(f"autocast dtype {dt} is not supported (the reference uses torch.bfloat16)")
            return "bf16"
        return "fp32"

    def _encode(self, input_ids, attention_mask, sparse):
        """sparse: False = dense head, True = sparse head, "both" = (sparse, dense) from one backbone pass."""
        if input_ids.dim() != 2 or attention_mask.shape != input_ids.shape:
            raise ValueError("input_ids and attention_mask must both be [batch, length]")
        prec = self.resolve_precision()
        if prec == "fp32" and self.fp32_planes == 0:
            raise _lib.SrHipError("fp32-regime encode requested (no torch.autocast(bf16) active) but the model was built "
                                  "with fp32_planes=0; wrap the call in torch.autocast('cuda', dtype=torch.bfloat16) or "
                                  "build the model with fp32_planes=3")
        self._ensure_engine(input_ids)
        src_device = input_ids.device
        ids = input_ids.to(device=self._device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self._device, dtype=torch.int64).contiguous()
        B, L = ids.shape
        if sparse == "both":
            out_s = torch.empty((B, self.config.vocab_size), dtype=torch.float32, device=self._device)
            out_d = torch.empty((B, self.config.hidden_size), dtype=torch.float32, device=self._device)
            with torch.cuda.device(self._device):
                stream = _lib.stream_ptr()
                for b0, b1 in self._call_ranges(mask, False):          # the dense span is the larger one
                    _lib.check(self._lib.sr_encode_both(self._h, ids[b0:b1].data_ptr(), mask[b0:b1].data_ptr(), b1 - b0, L,
                                                        int(prec == "fp32"), out_s[b0:b1].data_ptr(), out_d[b0:b1].data_ptr(),
                                                        stream), "sr_encode_both")
            return (out_s, out_d) if src_device == self._device else (out_s.to(src_device), out_d.to(src_device))
        width = self.config.vocab_size if sparse else self.config.hidden_size
        out = torch.empty((B, width), dtype=torch.float32, device=self._device)
        what = ("sr_encode_sparse" if sparse else "sr_encode_dense") + ("_fp32" if prec == "fp32" else "")
        fn = getattr(self._lib, what)
        with torch.cuda.device(self._device):
            stream = _lib.stream_ptr()
            for b0, b1 in self._call_ranges(mask, sparse):
                _lib.check(fn(self._h, ids[b0:b1].data_ptr(), mask[b0:b1].data_ptr(), b1 - b0, L,
                              out[b0:b1].data_ptr(), stream), what)
        return out if src_device == self._device else out.to(src_device)

    def _encode_many(self, batches, sparse):
        """Several collator batches in ONE pass of the engine (sr_encode_rows).  batches: [(input_ids [B_i, L_i], attention_mask)].
        The reference's drivers hand the encoder eval_batch_size rows per call (eval_dense.py:94-106, indexer.py:382-403; 128
        queries = ~1 100 tokens: 5 rows of 256-row GEMM tiles for 256 CUs); here the rows of all batches are laid right-aligned
        into one [B, L] matrix (a narrower batch gets extra LEFT padding) with a per-row shift, so every row keeps the
        position_ids it had in its own batch - its output is bit-identical to encoding that batch alone - and the engine
        cuts the rows by its token budget, not by the loader's batch size.  Returns the rows of all batches, in order."""
        if not batches:
            raise ValueError("encode_batches needs at least one batch")
        for ids, mask in batches:
            if ids.dim() != 2 or mask.shape != ids.shape:
                raise ValueError("input_ids and attention_mask must both be [batch, length]")
        prec = self.resolve_precision()
        if prec == "fp32" and self.fp32_planes == 0:
            raise _lib.SrHipError("fp32-regime encode requested (no torch.autocast(bf16) active) but the model was built with fp32_planes=0")
        self._ensure_engine(batches[0][0])
        src_device = batches[0][0].device
        B = sum(int(i.shape[0]) for i, _ in batches)
        L = max(int(i.shape[1]) for i, _ in batches)
        dev = self._device
        ids = torch.zeros((B, L), dtype=torch.int64, device=dev)
        mask = torch.zeros((B, L), dtype=torch.int64, device=dev)
        shift = torch.empty((B,), dtype=torch.int32, device=dev)
        b0 = 0
        for bi, bm in batches:
            n, l = int(bi.shape[0]), int(bi.shape[1])
            ids[b0:b0 + n, L - l:] = bi.to(device=dev, dtype=torch.int64)
            mask[b0:b0 + n, L - l:] = bm.to(device=dev, dtype=torch.int64)
            shift[b0:b0 + n] = L - l
            b0 += n
        mode = 2 if sparse == "both" else (1 if sparse else 0)
        out_s = torch.empty((B, self.config.vocab_size), dtype=torch.float32, device=dev) if mode != 0 else None
        out_d = torch.empty((B, self.config.hidden_size), dtype=torch.float32, device=dev) if mode != 1 else None
        with torch.cuda.device(dev):
            stream = _lib.stream_ptr()
            for r0, r1 in self._call_ranges(mask, sparse is True):
                _lib.check(self._lib.sr_encode_rows(self._h, ids[r0:r1].data_ptr(), mask[r0:r1].data_ptr(), r1 - r0, L,
                                                    shift[r0:r1].data_ptr(), mode, int(prec == "fp32"),
                                                    out_s[r0:r1].data_ptr() if out_s is not None else None,
                                                    out_d[r0:r1].data_ptr() if out_d is not None else None, stream), "sr_encode_rows")
        outs = tuple(o if src_device == dev else o.to(src_device) for o in (out_s, out_d) if o is not None)
        return outs if mode == 2 else outs[0]

    def _call_ranges(self, mask, sparse):
        """Row ranges [b0, b1) per C call.  The workspace holds max_batch_tokens PACKED tokens (pads are not computed), so
        a batch whose padded size fits goes through in one call; otherwise the rows are cut by the tokens they really
        pack to - the span plan_rows_kernel keeps per row (csrc/encoder.hip): for the dense head
        [min(first unmasked, L - len), L), for the sparse head [first unmasked, last unmasked]."""
        B, L = mask.shape
        if B <= self.max_batch_seqs and B * L <= self.max_batch_tokens:
            return [(0, B)]
        m = mask != 0
        n = m.sum(1)
        pos = torch.arange(L, device=mask.device)
        first = torch.where(m, pos, L).min(1).values
        if sparse:
            last = torch.where(m, pos, -1).max(1).values
            span = torch.where(n > 0, last + 1 - first, 0)
        else:
            span = torch.where(n > 0, L - torch.minimum(first, L - n), 0)
        span = span.cpu().tolist()                    # one small D2H per oversized batch
        out, b0, tok = [], 0, 0
        for b, t in enumerate(span):
            if t > self.max_batch_tokens:
                raise ValueError(f"row {b} packs to {t} tokens, the workspace holds {self.max_batch_tokens} (raise max_batch_tokens)")
            if b > b0 and (tok + t > self.max_batch_tokens or b - b0 >= self.max_batch_seqs):
                out.append((b0, b))
                b0, tok = b, 0
            tok += t
        out.append((b0, B))
        return out

    def last_hidden_state_packed(self):
        """fp32 [n_tokens, H] final-norm hidden states of the last encode call (test hook)."""
        n = ctypes.c_int64(0)
        buf = torch.empty((self.max_batch_tokens + 128, self.config.hidden_size), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.sr_model_last_hidden(self._h, buf.data_ptr(), buf.shape[0], ctypes.byref(n),
                                                      _lib.stream_ptr()), "sr_model_last_hidden")
        return buf[:n.value]

    def __del__(self):
        try:
            if self._h is not None and self._lib is not None:
                self._lib.sr_model_destroy(self._h)
                self._h = None
        except Exception:
            pass


# --------------------------------------------------------------------------- retrievers
class LLM2Retriever(torch.nn.Module):
    """llm_encoder.py:14-150 (inference surface)."""
    TRANSFORMER_CLS = None      # name of the reference's backbone class (checked against adapter auto_mapping)
    TARGET_MODULES = TARGET_MODULES
    HAS_LM_HEAD = False
    _tied_weights_keys = None

    def __init__(self, base_model):
        super().__init__()
        self.base_model = base_model

    def encode(self, **inputs):
        raise NotImplementedError

    def doc_encode(self, **inputs):
        return self.encode(**inputs)

    def query_encode(self, **inputs):
        return self.encode(**inputs)

    _HEAD = None        # False dense, True sparse, "both": which head(s) encode() returns

    def encode_batches(self, batches):
        """MI355X-side extension of the reference API: `encode` over SEVERAL collator batches in one pass of the engine.
        batches: a list of dicts with "input_ids" / "attention_mask" [B_i, L_i] (other keys ignored).  Returns what
        torch.cat([self.encode(**b) for b in batches]) returns - the same bits, row for row (every row keeps the positions it
        has in its own batch) - but the rows are cut by the engine's token budget instead of the loader's batch size."""
        return self.base_model._encode_many([(b["input_ids"], b["attention_mask"]) for b in batches], self._HEAD)

    def forward(self, **inputs):
        raise NotImplementedError("training losses are out of scope of the MI355X inference path")

    def to(self, *args, **kwargs):
        self.base_model.to(*args, **kwargs)
        return self

    def cuda(self, device=None):
        self.base_model.cuda(device)
        return self

    # ---- loaders -------------------------------------------------------------------
    @classmethod
    def _make(cls, base_model, **kw):
        return cls(base_model, **kw)

    @classmethod
    def _check_adapter_layout(cls, state, lora_config):
        return  # LLM2Retriever.load does no layout check (llm_encoder.py:105-129)

    @classmethod
    def _load_impl(cls, model_name_or_path, lora_name_or_path, merge_peft, is_trainable, access_token, **make_kw):
        if is_trainable or not merge_peft:
            raise NotImplementedError("only merged, inference-only adapters are supported on the HIP path")
        lora = None
        if lora_name_or_path is not None:
            state = _load_adapter_state(lora_name_or_path)
            ldir = lora_name_or_path if os.path.isdir(lora_name_or_path) else None
            if ldir is None:
                from huggingface_hub import hf_hub_download
                cfg_path = hf_hub_download(lora_name_or_path, "adapter_config.json")
            else:
                cfg_path = os.path.join(ldir, "adapter_config.json")
            with open(cfg_path) as f:
                lora_config = json.load(f)
            cls._check_adapter_layout(state, lora_config)
            # what peft's merge_and_unload would do that this loader does not implement: refuse instead of ignoring
            if lora_config.get("use_dora"):
                raise NotImplementedError("DoRA adapters (use_dora) are not supported")
            if lora_config.get("rank_pattern") or lora_config.get("alpha_pattern"):
                raise NotImplementedError("per-module rank_pattern / alpha_pattern are not supported")
            if lora_config.get("bias", "none") not in ("none", None):
                raise NotImplementedError(f"adapter bias='{lora_config['bias']}' is not supported (Llama linears have no bias)")
            r, alpha = int(lora_config["r"]), float(lora_config["lora_alpha"])
            scale = alpha / (r ** 0.5) if lora_config.get("use_rslora") else alpha / r
            A, Bm, extra = {}, {}, {}
            for k, v in state.items():
                name, ab = _adapter_target(k)
                if name is not None:
                    (A if ab == "A" else Bm)[name] = v
                    continue
                # A module listed in modules_to_save (e.g. a trained lm_head or embed_tokens, --lora_modules_to_save) ships
                # as a whole tensor: peft saves it as `base_model.model.<module>.weight` (adapter name and the
                # `modules_to_save.` level stripped), older files keep `.modules_to_save.<adapter>.`; PeftModel.from_pretrained
                # + merge_and_unload puts it in place of the base tensor.
                kk = k[len("base_model.model."):] if k.startswith("base_model.model.") else k
                if ".original_module." in kk:
                    continue                                   # peft's frozen copy of the base tensor
                for tag in (".modules_to_save.default.", ".modules_to_save."):
                    if tag in kk:
                        kk = kk.replace(tag, ".")
                        break
                if any(t in kk for t in ("lora_embedding_", "lora_magnitude_vector", ".lora_")) or not kk.endswith(".weight"):
                    raise ValueError(f"adapter tensor '{k}' is neither a LoRA A/B pair of a linear nor a full replacement weight; "
                                     "loading would silently drop it")
                extra[_canonical_name(kk)] = v
            if set(A) != set(Bm):
                raise ValueError("adapter has unmatched lora_A / lora_B tensors")
            lora = {"scale": scale, "A": A, "B": Bm, "extra": extra}
        base_dir = _resolve_dir(model_name_or_path, access_token)
        with open(os.path.join(base_dir, "config.json")) as f:
            config = LlamaConfigLite.from_dict(json.load(f))
        weights = _read_checkpoint(base_dir)
        if lora is not None:
            extra = lora.pop("extra")
            unknown = [n for n in extra if n not in weights and not (n == "lm_head.weight" and cls.HAS_LM_HEAD)]
            if unknown:
                raise ValueError(f"adapter replaces tensors absent from the base checkpoint: {unknown[:3]}")
            if "lm_head.weight" in extra and config.tie_word_embeddings:
                config.tie_word_embeddings = False             # the trained head no longer equals embed_tokens
            weights.update(extra)
            missing = [n for n in lora["A"] if n not in weights]
            if missing:
                raise ValueError(f"adapter targets tensors absent from the base checkpoint: {missing[:3]}")
        if cls.HAS_LM_HEAD and not config.tie_word_embeddings and "lm_head.weight" not in weights:
            raise ValueError("checkpoint has no lm_head.weight and tie_word_embeddings is false")
        backbone = HipLlamaBackbone(config, weights, has_lm_head=cls.HAS_LM_HEAD, lora=lora)
        return cls._make(backbone, **make_kw)

    @classmethod
    def load(cls, model_name_or_path, lora_name_or_path=None, merge_peft=True, is_trainable=False, access_token=None):
        return cls._load_impl(model_name_or_path, lora_name_or_path, merge_peft, is_trainable, access_token)

    @classmethod
    def load_from_lora(cls, lora_name_or_path, merge_peft=True, is_trainable=False, access_token=None):
        """llm_encoder.py:131-150: adapter_config.json names the base model."""
        if os.path.isdir(lora_name_or_path):
            adapter_config_path = os.path.join(lora_name_or_path, "adapter_config.json")
        else:
            from huggingface_hub import hf_hub_download
            adapter_config_path = hf_hub_download(lora_name_or_path, "adapter_config.json")
        with open(adapter_config_path, "r") as f:
            adapter_config = json.load(f)
        return cls.load(adapter_config["base_model_name_or_path"], lora_name_or_path=lora_name_or_path,
                        merge_peft=merge_peft, is_trainable=is_trainable, access_token=access_token)

    @classmethod
    def from_weights(cls, config, weights, max_batch_tokens=32768, max_batch_seqs=1024, precision="auto", fp32_planes=None,
                     **make_kw):
        """Build from an in-memory state dict (HF Llama names -> numpy/torch tensors)."""
        weights = {_canonical_name(k): v for k, v in weights.items()}
        backbone = HipLlamaBackbone(config, weights, has_lm_head=cls.HAS_LM_HEAD, max_batch_tokens=max_batch_tokens,
                                    max_batch_seqs=max_batch_seqs, precision=precision, fp32_planes=fp32_planes)
        return cls._make(backbone, **make_kw)

    def save_pretrained(self, save_dir):
        raise NotImplementedError("saving is out of scope of the MI355X inference path")


class DecoderOnlyBiSparse(LLM2Retriever):
    """llm_encoder.py:175-196."""
    HAS_LM_HEAD = True
    _HEAD = True

    def __init__(self, base_model):
        super().__init__(base_model)
        self.vocab_size = self.base_model.config.vocab_size

    def encode(self, **inputs):
        return self.base_model._encode(inputs["input_ids"], inputs["attention_mask"], sparse=True)

    def rerank_forward(self, **inputs):
        query_reps = self.encode(**inputs["tokenized_queries"])
        doc_reps = self.encode(**inputs["tokenized_docs"])
        return (query_reps * doc_reps).sum(dim=-1)


class DecoderOnlyBiDense(LLM2Retriever):
    """llm_encoder.py:370-520."""
    HAS_LM_HEAD = False
    _HEAD = False

    def __init__(self, base_model, T=0.01):
        super().__init__(base_model)
        self.hidden_size = self.base_model.config.hidden_size
        self.T = T

    def encode(self, **inputs):
        return self.base_model._encode(inputs["input_ids"], inputs["attention_mask"], sparse=False)

    def rerank_forward(self, **inputs):
        query_reps = self.encode(**inputs["tokenized_queries"])
        doc_reps = self.encode(**inputs["tokenized_docs"])
        return (query_reps * doc_reps).sum(dim=-1)

    @classmethod
    def _check_adapter_layout(cls, state, lora_config):
        # llm_encoder.py:494-495 and :512-514
        first = list(state.keys())[0]
        assert "base_model.model.model.layers" not in first
        assert "base_model.model.layers" in first
        am = lora_config.get("auto_mapping") or {}
        assert am.get("base_model_class") == cls.TRANSFORMER_CLS, (am.get("base_model_class"), cls.TRANSFORMER_CLS)

    @classmethod
    def load(cls, model_name_or_path, lora_name_or_path=None, merge_peft=True, is_trainable=False, T=0.01,
             access_token=None):
        return cls._load_impl(model_name_or_path, lora_name_or_path, merge_peft, is_trainable, access_token, T=T)


class LlamaBiSparse(DecoderOnlyBiSparse):
    TRANSFORMER_CLS = "LlamaBiForMNTP"


class DecoderOnlyBiHybrid(DecoderOnlyBiSparse):
    """The model HybridIndexer / HybridRetriever drive (/root/reference/scaling_retriever/indexer.py:710-1019:
    `batch_sparse_reps, batch_dense_reps = self.model.encode(**inputs)`, `self.model.hidden_size`): a LlamaBiForMNTP
    backbone whose ONE forward pass feeds both heads - the sparse head of llm_encoder.py:186-196 and the dense head of
    :424-443.  (The reference ships the retriever classes but no encoder class of this shape; eval_reranker.py:120 names a
    LlamaBiHybridRetrieverForNCE that is not in the tree.)"""

    _HEAD = "both"

    def __init__(self, base_model, T=0.01):
        super().__init__(base_model)
        self.hidden_size = self.base_model.config.hidden_size
        self.T = T

    def encode(self, **inputs):
        return self.base_model._encode(inputs["input_ids"], inputs["attention_mask"], sparse="both")

    def rerank_forward(self, **inputs):
        raise NotImplementedError


class LlamaBiHybrid(DecoderOnlyBiHybrid):
    TRANSFORMER_CLS = "LlamaBiForMNTP"


class LlamaBiDense(DecoderOnlyBiDense):
    TRANSFORMER_CLS = "LlamaBiModel"


LlamaBiSparseForNCE = LlamaBiSparse
LlamaBiDenseForNCE = LlamaBiDense
