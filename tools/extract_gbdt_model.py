#!/usr/bin/env python3
"""Convert the learned-ANI regression model parameters into a flat binary table.

The reference embeds two trained gradient-boosted-tree models as JSON text
(/root/reference/src/model.rs:2 = c125 model, :5 = c200 model; consumed by regression.rs:12-28).
The numbers are trained *weights* (data), needed for output parity because learned ANI is
default-on at c >= 70 (parse.rs:885-889).  This script (run in the build container only) parses
the JSON and writes, per model, skani_amd/data/gbdt_c{125,200}.bin:

    u32 magic 'GBDT' | u32 n_trees | u32 n_features | f32 shrinkage | f32 bias | u32 n_nodes_total
    u32 tree_offset[n_trees+1]
    node[n_nodes_total] = { i32 feature (-1 = leaf) ; f32 threshold ; f32 pred ; i32 left ; i32 right }
                          (left/right are node indices local to the tree)

f32 values are obtained the way serde_json fills an f32 field: decimal -> f64 -> f32.
"""
import json, os, re, struct, sys
import numpy as np

SRC = "/root/reference/src/model.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "skani_amd", "data")


def convert(js, path):
    m = json.loads(js)
    conf = m["conf"]
    assert conf["loss"] == "LAD" and not conf["initial_guess_enabled"]
    trees = m["trees"]
    assert len(trees) == conf["iterations"]
    offs = [0]; nodes = []
    for t in trees:
        tn = t["tree"]["tree"]
        for i, nd in enumerate(tn):
            assert nd["index"] == i
            v = nd["value"]
            assert v["missing"] == 0
            if v["is_leaf"]:
                nodes.append((-1, 0.0, v["pred"], 0, 0))
            else:
                nodes.append((v["feature_index"], v["feature_value"], v["pred"], nd["left"], nd["right"]))
        offs.append(len(nodes))
    with open(path, "wb") as f:
        f.write(struct.pack("<4sIIffI", b"GBDT", len(trees), conf["feature_size"],
                            float(np.float32(conf["shrinkage"])), float(np.float32(m["bias"])), len(nodes)))
        f.write(np.array(offs, np.uint32).tobytes())
        for (fi, thr, pred, l, r) in nodes:
            f.write(struct.pack("<iffii", fi, float(np.float32(thr)), float(np.float32(pred)), l, r))
    print(path, "trees", len(trees), "nodes", len(nodes), "shrinkage", conf["shrinkage"], "bias", m["bias"],
          "keys", sorted(m.keys()))


def main():
    txt = open(SRC).read()
    blobs = re.findall(r'r#"\s*(\{.*?\})\s*"#', txt, re.S)
    assert len(blobs) == 2, len(blobs)
    names = re.findall(r"pub const (\w+)\s*:", txt)
    print(names)
    os.makedirs(OUT, exist_ok=True)
    for name, js in zip(names, blobs):
        tag = "c200" if "200" in name else "c125"
        convert(js, os.path.join(OUT, f"gbdt_{tag}.bin"))


if __name__ == "__main__":
    main()
