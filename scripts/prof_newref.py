import cProfile, pstats, sys, os, shutil, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from wisecondorx_amd import main as cli, npz_io, synth
wd="/tmp/wcx_prof"; shutil.rmtree(wd, ignore_errors=True); os.makedirs(wd)
co = synth.Cohort(15000, female_y=0.1)
samples, genders = co.cohort(int(sys.argv[1]))
files=[]
for i,s in enumerate(samples):
    f=os.path.join(wd,"s%03d.npz"%i); npz_io.save_sample(f,s,15000); files.append(f)
random.seed(1)
pr=cProfile.Profile(); pr.enable()
cli.main(["--loglevel","warning","newref"]+files+[os.path.join(wd,"ref.npz"),"--binsize","15000","--yfrac","0.004"])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
shutil.rmtree(wd, ignore_errors=True)
