import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], "value ms", d.get("ms_per_step"), "cfg3", d.get("configs3",{}).get("ms_per_step"), "cfg4", d.get("configs4",{}).get("ms_per_step"))
