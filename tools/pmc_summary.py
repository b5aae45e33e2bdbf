# Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel.
import sys, glob
import pandas as pd
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        d = pd.read_csv(f)
        d["k"] = d["Kernel_Name"].str.extract(r"(k_\w+)")
        g = d.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
        pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
        print(path); print(g.round(0).to_string())
