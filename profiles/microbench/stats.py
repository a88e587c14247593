import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:110].ljust(110), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1000)).rjust(9), 'us', r['Percentage'])
