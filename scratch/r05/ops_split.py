import csv,sys
rows=list(csv.DictReader(open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/ops_prof/k_kernel_trace.csv')))
seq=[(r['Kernel_Name'], int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows]
for n in ['det_decode_sort','nms_mask','nms_scan','target_match','target_rows','target_write']:
    d=[t for k,t in seq if n in k]
    print(n,len(d), [round(sum(d[i:i+21])/len(d[i:i+21])/1e3,1) for i in range(0,len(d),21)])
