# the analysis_entry leg of a bench line on stdin, in one line:  python bench.py ... | python tools/exp/print_analysis.py
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])["analysis_entry"]
print(d["ms"], round(d["ms_per_entry"], 3))
