"""VMAF model files (models/vmaf_*.json, loaded unchanged) -> per-frame scores on the host.

Stands where libvmaf's predict.c / svm.cpp / read_json_model.c stand behind the reference's
`model=version=<name>` option (app/vmaf_analyzer.py:377; models listed at
app/ui/tabs/analysis_tab.py:1034-1046).  nu-SVR with an RBF kernel, features linearly rescaled by
`slopes`/`intercepts`, prediction de-normalised with entry 0, clipped to `score_clip`;
`score_transform` is applied only on request, as in libvmaf.
"""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass, field

import numpy as np

MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
BUILTIN_VERSIONS = ("vmaf_v0.6.1", "vmaf_4k_v0.6.1", "vmaf_b_v0.6.3", "vmaf_v0.6.1neg", "vmaf_4k_v0.6.1neg",
                    "vmaf_float_v0.6.1", "vmaf_float_4k_v0.6.1", "vmaf_float_v0.6.1neg", "vmaf_float_b_v0.6.3")


@dataclass
class SvmModel:
    feature_names: list          # model order, e.g. VMAF_integer_feature_adm2_score
    slopes: np.ndarray           # [1 + n_features]
    intercepts: np.ndarray
    sv: np.ndarray               # [n_sv, n_features]
    coef: np.ndarray             # [n_sv]
    gamma: float
    rho: float
    score_clip: tuple | None
    score_transform: dict | None
    feature_opts: list = field(default_factory=list)

    @property
    def metric_keys(self):
        """libvmaf log keys of the model's features: VMAF_integer_feature_adm2_score -> integer_adm2."""
        return [feature_key(n) for n in self.feature_names]

    def predict(self, feats: np.ndarray, enable_transform: bool = False) -> np.ndarray:
        """feats [n, n_features] in model order -> scores [n]."""
        feats = np.asarray(feats, np.float64)
        n = feats.shape[0]
        block = 512   # fixed cache-sized blocks: a frame's score does not depend on how many frames ride along
        if n <= block:
            return self._predict_block(feats, enable_transform)
        return np.concatenate([self._predict_block(feats[s0:s0 + block], enable_transform) for s0 in range(0, n, block)])

    def _predict_block(self, feats: np.ndarray, enable_transform: bool = False) -> np.ndarray:
        x = np.asarray(feats, np.float64) * self.slopes[1:] + self.intercepts[1:]
        # ||x - sv||^2 = ||x||^2 + ||sv||^2 - 2 x.sv : one small GEMM instead of an [n, n_sv, 6] temporary
        # (f64; features and SVs are O(1), so the cancellation costs ~1e-15 -- far below libvmaf's %.6f)
        d2 = x @ self.sv.T
        d2 *= -2.0
        d2 += (x * x).sum(1)[:, None]
        d2 += (self.sv * self.sv).sum(1)[None, :]
        np.maximum(d2, 0.0, out=d2)
        d2 *= -self.gamma
        np.exp(d2, out=d2)
        y = d2 @ self.coef - self.rho
        y = (y - self.intercepts[0]) / self.slopes[0]
        if enable_transform and self.score_transform:
            t = self.score_transform
            z = np.zeros_like(y)
            for k, pw in (("p0", 0), ("p1", 1), ("p2", 2)):
                if k in t:
                    z = z + float(t[k]) * y ** pw
            if str(t.get("out_gte_in", "false")).lower() == "true":
                z = np.maximum(z, y)
            if str(t.get("out_lte_in", "false")).lower() == "true":
                z = np.minimum(z, y)
            y = z
        if self.score_clip is not None:
            y = np.clip(y, self.score_clip[0], self.score_clip[1])
        return y


@dataclass
class VmafModel:
    name: str
    path: str
    models: list                 # [SvmModel]; >1 for BOOTSTRAP collections (entry 0 = full model)

    @property
    def main(self) -> SvmModel:
        return self.models[0]

    @property
    def is_integer(self) -> bool:
        return any("integer" in n for n in self.main.feature_names)

    @property
    def vif_border(self) -> int:
        """pqa_config.vif_border for this model: VMAF_integer_feature_vif_* names integer_vif.c, whose padding is
        reflect-101; VMAF_feature_vif_* (vmaf_float_*) names float_vif / vif_tools.c (include/pqa_vmaf.h)."""
        return 1 if self.is_integer else 0

    @property
    def vif_enhn_gain_limit(self) -> float:
        return _opt(self.main, "vif_enhn_gain_limit")

    @property
    def adm_enhn_gain_limit(self) -> float:
        return _opt(self.main, "adm_enhn_gain_limit")


def _opt(m: SvmModel, key: str) -> float:
    for d in m.feature_opts or []:
        if isinstance(d, dict) and key in d:
            return float(d[key])
    return 100.0


def feature_key(model_feature_name: str) -> str:
    m = re.match(r"VMAF_(integer_)?feature_(.*)_score$", model_feature_name)
    if not m:
        return model_feature_name
    return (m.group(1) or "") + m.group(2)


def _parse_libsvm(text: str, n_features: int):
    lines = [ln for ln in text.strip().split("\n")]
    hdr, i = {}, 0
    while lines[i].strip() != "SV":
        parts = lines[i].split()
        hdr[parts[0]] = parts[1:]
        i += 1
    if hdr.get("svm_type", [""])[0] != "nu_svr" or hdr.get("kernel_type", [""])[0] != "rbf":
        raise ValueError("only nu_svr / rbf libsvm models are supported")
    rows = [ln.split() for ln in lines[i + 1:] if ln.strip()]
    coef = np.array([float(r[0]) for r in rows])
    sv = np.zeros((len(rows), n_features))
    for j, r in enumerate(rows):
        for tok in r[1:]:
            idx, val = tok.split(":")
            sv[j, int(idx) - 1] = float(val)
    return sv, coef, float(hdr["gamma"][0]), float(hdr["rho"][0])


def _svm_from_dict(md: dict) -> SvmModel:
    if md.get("norm_type", "linear_rescale") != "linear_rescale":
        raise ValueError(f"unsupported norm_type {md.get('norm_type')}")
    names = list(md["feature_names"])
    sv, coef, gamma, rho = _parse_libsvm(md["model"], len(names))
    clip = md.get("score_clip")
    return SvmModel(names, np.asarray(md["slopes"], np.float64), np.asarray(md["intercepts"], np.float64),
                    sv, coef, gamma, rho, tuple(clip) if clip else None, md.get("score_transform"),
                    md.get("feature_opts_dicts") or [])


def resolve_model_path(model: str | None) -> str:
    """`model=version=<name>` -> bundled JSON; anything with a path separator / existing file -> that file
    (the same split the reference makes at app/vmaf_analyzer.py:335-341,377)."""
    if not model:
        model = "vmaf_v0.6.1"
    if model.startswith("path="):
        model = model[5:]
    if os.sep in model or "/" in model or "\\" in model or os.path.isfile(model):
        return model
    name = model[:-5] if model.endswith(".json") else model
    cand = os.path.join(MODELS_DIR, name + ".json")
    if not os.path.isfile(cand):
        raise FileNotFoundError(f"unknown VMAF model '{model}' (no {cand})")
    return cand


def load_model(model: str | None = "vmaf_v0.6.1") -> VmafModel:
    path = resolve_model_path(model)
    with open(path, "r") as f:
        data = json.load(f)
    name = os.path.splitext(os.path.basename(path))[0]
    if "model_dict" in data:
        return VmafModel(name, path, [_svm_from_dict(data["model_dict"])])
    keys = sorted((k for k in data if k.isdigit()), key=int)   # BOOTSTRAP_LIBSVMNUSVR collection
    if not keys:
        raise ValueError(f"{path}: neither a model_dict nor a bootstrap collection")
    return VmafModel(name, path, [_svm_from_dict(data[k]["model_dict"]) for k in keys])


# ---------------------------------------------------------------------------------------------
# feature records -> libvmaf metric columns
# ---------------------------------------------------------------------------------------------
def metrics_from_records(rec: np.ndarray, width: int, height: int, prefix: str = "") -> dict:
    """[n,24] engine records -> ordered dict of per-frame metric arrays named as libvmaf logs them
    (prefix 'integer_' for the default models).  Epilogues of float_adm.c / float_vif.c / float_motion.c."""
    rec = np.asarray(rec, np.float64)
    n = rec.shape[0]
    out = {}
    motion = rec[:, 16].copy()
    m2 = motion.copy()
    if n > 1:
        m2[:-1] = np.minimum(motion[:-1], motion[1:])
    out[prefix + "motion2"] = m2
    out[prefix + "motion"] = motion
    num, den = rec[:, 8:12], rec[:, 12:16]
    limit = 1e-10 * (width * height) / (1920.0 * 1080.0)
    ns, ds = num.sum(1), den.sum(1)
    ns = np.where(ns < limit, 0.0, ns)
    ds = np.where(ds < limit, 0.0, ds)
    with np.errstate(divide="ignore", invalid="ignore"):
        out[prefix + "adm2"] = np.where(ds == 0.0, 1.0, ns / np.where(ds == 0.0, 1.0, ds))
        for s in range(4):
            out[prefix + f"adm_scale{s}"] = num[:, s] / den[:, s]
        for s in range(4):
            out[prefix + f"vif_scale{s}"] = rec[:, s] / rec[:, 4 + s]
    return out


def score_frames(model: VmafModel, metrics: dict, enable_transform: bool = False) -> dict:
    """Adds 'vmaf' (and bootstrap statistics for collections) to a copy of `metrics`."""
    main = model.main
    cols = []
    for key in main.metric_keys:
        if key not in metrics:
            alt = key.replace("integer_", "")
            if alt not in metrics:
                raise KeyError(f"model feature {key} was not extracted")
            key = alt
        cols.append(np.asarray(metrics[key], np.float64))
    X = np.stack(cols, 1) if cols else np.zeros((0, 0))
    out = dict(metrics)
    out["vmaf"] = main.predict(X, enable_transform)
    if len(model.models) > 1:
        # BOOTSTRAP collections (vmaf_b_v0.6.3): entry 0 scores `vmaf`, the other 20 give the spread.  PARITY UNPINNED:
        # the four statistic NAMES follow libvmaf's log keys as remembered (vmaf_bagging, vmaf_stddev, vmaf_ci_p95_lo/hi);
        # whether libvmaf uses the population std (ddof 0, as here) and linear-interpolated 2.5 / 97.5 percentiles is a
        # VERIFY item -- the reference holds no bootstrap output to check against.
        boots = np.stack([m.predict(X, enable_transform) for m in model.models[1:]], 0)
        out["vmaf_bagging"] = boots.mean(0)
        out["vmaf_stddev"] = boots.std(0)
        out["vmaf_ci_p95_lo"] = np.percentile(boots, 2.5, axis=0)
        out["vmaf_ci_p95_hi"] = np.percentile(boots, 97.5, axis=0)
    return out


def pool(values: np.ndarray) -> dict:
    v = np.asarray(values, np.float64)
    if v.size == 0:
        return {"min": 0.0, "max": 0.0, "mean": 0.0, "harmonic_mean": 0.0}
    return {"min": float(v.min()), "max": float(v.max()), "mean": float(v.mean()),
            "harmonic_mean": float(1.0 / np.mean(1.0 / (v + 1.0)) - 1.0)}
