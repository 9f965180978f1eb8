import re,sys
# parse "[rt] timeline" lines: span, sum_end>>8, sum_exh>>8, start ; waves given
n=int(sys.argv[1])
for line in sys.stdin:
    m=re.search(r"span (\d+), sum_end (\d+), sum_exhaust (\d+), start (\d+)", line)
    if m:
        span,se,sx,st=map(int,m.groups())
        mean_end=(se/n - st)/span; mean_exh=(sx/n - st)/span
        print("span ticks(10ns) %d: mean wave end at %.3f, mean exhaustion at %.3f of the kernel span"%(span,mean_end,mean_exh))
