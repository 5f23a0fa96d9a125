#!/usr/bin/env python3
"""Parse the reference's shipped YAML configs with fatezero_amd.config_driver and commit the per-prompt call plans of the
BASELINE configurations as fixtures (tests/fixtures/plan_*.json): /root/reference does not exist on the GPU box, the plans
do.  Run in the authoring container:  python scripts/gen_plan_fixtures.py"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fatezero_amd import config_driver as CD  # noqa: E402

REF = "/root/reference/config"
PICK = {"cfg1": "low_resource_teaser/jeep_watercolor_ddim_10_steps.yaml", "cfg2": "teaser/jeep_posche.yaml",
        "cfg3": "style/sun_flower_van_gogh.yaml", "cfg4": "attribute/squ_carrot_robot_eggplant.yaml",
        "cfg5": "shape/swan_duck_flamingo.yaml", "latent_blend": "teaser/jeep_posche_local_latent_blend.yaml"}


def summary(path):
    cfg = CD.load_config(path)
    out = {"file": os.path.relpath(path, "/root/reference"), "unresolved": cfg.unresolved,
           "model_config": cfg.get("model_config"), "dataset_config": cfg.get("dataset_config")}
    if "editing_config" in cfg:
        ed = cfg["editing_config"]
        out["num_inference_steps"] = ed.get("num_inference_steps")
        out["use_inversion_attention"] = ed.get("use_inversion_attention")
        out["plan"] = CD.plan_edits(ed, cfg.get("dataset_config", {}).get("prompt"))
    return out


if __name__ == "__main__":
    files = sorted(glob.glob(os.path.join(REF, "*", "*.yaml")))
    index = {}
    for f in files:
        s = summary(f)
        index[s["file"]] = {"unresolved": s["unresolved"], "n_calls": len(s.get("plan", []))}
    os.makedirs(os.path.join(ROOT, "tests", "fixtures"), exist_ok=True)
    json.dump(index, open(os.path.join(ROOT, "tests", "fixtures", "plan_index.json"), "w"), indent=1, sort_keys=True)
    for name, rel in PICK.items():
        json.dump(summary(os.path.join(REF, rel)), open(os.path.join(ROOT, "tests", "fixtures", f"plan_{name}.json"), "w"),
                  indent=1, sort_keys=True)
    print(f"{len(files)} configs parsed; fixtures written for {sorted(PICK)}")
