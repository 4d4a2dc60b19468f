import csv,glob,sys
for d in sys.argv[1:]:
    f=glob.glob(d+"/stats/*/*kernel_trace.csv")[0]
    tr=list(csv.DictReader(open(f)))
    ev=[(r["Kernel_Name"].split("(")[0].replace("void ","")[:24], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X")) for r in tr if "k_firth" in r["Kernel_Name"] or "k_glm" in r["Kernel_Name"]]
    half=len(ev)//2
    print(d, "n", len(ev))
    for e in ev[half:half+16]: print("   ", e)
    print("   total ms second half", sum(e[1] for e in ev[half:]))
