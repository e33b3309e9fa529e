"""Print the error objects (if any) of a bench.py output file, the judged line's length and a few headline scalars."""
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("SECONDARY ") or l.startswith("{")]
sec = json.loads(lines[0][len("SECONDARY "):]) if lines[0].startswith("SECONDARY ") else {}
main = json.loads(lines[-1])


def errs(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == "error":
                print("ERROR at", path, v)
            errs(v, path + "/" + k)


errs(sec)
r = main["roofline"]
print("line", len(lines[-1]), "value", main["value"], "split", r.get("split_bf16_value"), r.get("split_bf16_frac_of_bf16_peak"), "hooked", r.get("hooked_loop_value"),
      "write roof", r.get("reparam_frac_of_write_roof"), "train", r.get("training_step_ms"), r.get("training_step_frac"))
if sec:
    print("split other configs", json.dumps(sec["split_bf16"].get("other_configs")))
    print("configs[1] parity", json.dumps(sec["configs"]["configs[1]"].get("parity")))
    print({k: (v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic")) for k, v in sec["configs"].items()})
    print("fusion_ab", json.dumps(sec["split_bf16"].get("fusion_ab"))[:600])
