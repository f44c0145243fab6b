import json,sys
d=json.load(open(sys.argv[1]))
c=d["cmax"]; print({k:c[k] for k in c if k.endswith("_ms")}); print({k:c["c4"][k] for k in c["c4"] if k.endswith("_ms")}); print(c["c4"]["bfgs_numeric_grads(reference default)"]["seconds"], c["c4"]["bfgs_analytic_consistent_grad"]["seconds"])
