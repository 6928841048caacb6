import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sinks" in r["Name"]: print(sys.argv[1], r["Name"][:40], round(float(r["AverageNs"])/1000,1))
