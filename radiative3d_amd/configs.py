"""Run configurations as the reference's do-*.sh scripts assemble them
(scripts/do-fundamentals.sh:396-419), parametrised by TOA degree so tests can
use small tables.  Values are the scripts' own (do-halfspace.sh:41-101,
do-crustpinch.sh:26-74, do-lopnor.sh:26-104 with event=expl,
do-spherical.sh:27-74)."""


def halfspace(deg=9, one_receiver=False):
    seis = ("--seis-p2p=0,0,0,260,0,0,2.737,0.105,10.0,1" if one_receiver else
            "--seis-p2p=0,0,0,183.85,183.85,0,2.737,0.105,10.0,48 "
            "--seis-p2p=0,0,0,260,0,0,2.737,0.105,10.0,48 "
            "--seis-p2p=0,0,0,240.21,-99.5,0,2.737,0.105,10.0,48")
    return ("--grid-compiled=40 "
            "--model-args=0.8,0.01,1.0,0.5,1000,0.8,0.01,1.0,0.5,1000,6.40,3.63,2.83,-60,6.40,3.63,2.83,-400 "
            "--range=900 --source=SDR,0,90,0,0.0 --source-loc=0,0,-5 --frequency=2.0 --timetolive=200 "
            f"--binsize=0.50 --toa-degree={deg} " + seis).split()


def crustpinch(deg=9):
    return ("--grid-compiled=5 "
            "--model-args=0.8,0.01,0.20,0.2,200,0.8,0.01,0.20,0.3,1500,0.8,0.01,0.20,0.3,1500,"
            "0.8,0.01,0.20,0.4,1500,0.8,0.01,0.20,0.5,900,2.0,30.0,5.0,.3666667,.4736842,1,1 "
            "--source=SDR,22.5,90,0 --source-loc=0,0,-10 --frequency=2.0 --timetolive=600 "
            f"--binsize=2.00 --toa-degree={deg} "
            "--seis-p2p=0,67.5,0,950,67.5,0,1.0,2.0,40.0,160 "
            "--seis-p2p=0,112.5,0,950,112.5,0,1.0,2.0,40.0,160 "
            "--seis-p2p=0,90,0,950,90,0,1.0,2.0,40.0,160").split()


def lopnor(deg=9, event="expl"):
    """do-lopnor.sh:26-104; event: the script's own switch (:31) -- "expl", the generic explosion
    2 km below Lop Nor (BASELINE config 3), or "eq", the Xinjiang earthquake 32 km down (what the
    script ships with)."""
    source = {"expl": "--source=EXPL --source-loc=425.54,-169.53,-1.02",
              "eq": "--source=SDR,125,40,90,0.0 --source-loc=425.54,-169.53,-31.02"}[event]
    return ("--grid-compiled=1 --flatten --range=1200 "
            "--model-args=0.8,0.01,0.5,0.2,50,0.8,0.01,0.5,0.3,1000,0.8,0.01,0.7,0.5,300 "
            f"{source} --frequency=2.0 --timetolive=600 "
            f"--binsize=2.00 --toa-degree={deg} "
            "--seis-p2p=425.54,-169.53,0.98,-390.04,-167.18,1.457,1.0,2.0,40.0,160 "
            "--seis-p2p=425.54,-169.53,0.98,-102.27,430.84,0.60,1.0,2.0,40.0,160").split()


def sphere(deg=9, source_depth=-10):
    """do-spherical.sh:27-74.  (The script passes the crust-pinch run's five scattering groups as
    --model-args; the whole-Earth model ignores them, do-spherical.sh:46, user_SphereEarth_inc.cpp:22-27.)"""
    return ("--grid-compiled=16 --source=SDR,22.5,90,0 "
            "--model-args=0.8,0.01,0.20,0.2,200,0.8,0.01,0.20,0.3,1500,0.8,0.01,0.20,0.3,1500,"
            "0.8,0.01,0.20,0.4,1500,0.8,0.01,0.20,0.5,900 "
            f"--source-loc=0,0,{source_depth} --frequency=2.0 --timetolive=8000 "
            f"--binsize=20.0 --toa-degree={deg} "
            "--seis-p2p=0,67.5,0,12000,67.5,0,20.0,20.0,400.0,160 "
            "--seis-p2p=0,112.5,0,12000,112.5,0,20.0,20.0,400.0,160 "
            "--seis-p2p=0,90,0,12000,90,0,20.0,20.0,400.0,160").split()


def sphere_deep(deg=9):
    """BASELINE config 4: the spherical-Earth run with the double-couple source 600 km deep."""
    return sphere(deg, source_depth=-600)


def crustpinch_vids(deg=9):
    """do-crustpinch-vids.sh:22-72: the crust-pinch model as a scatter-event video run -- pinned
    mean free paths, scattering without deflection (scatter events become dense check-points,
    ~10 SCT + REF events per history), 350 s, coarse bins, three 16-receiver lines.  With a
    dense event grid attached this is BASELINE config 5.  (The script names no model: an unspecified
    grid source is selection 0, which user.cpp:63-67 turns into 5, the crust pinch.)"""
    return ("--model-args=0.8,0.01,0.20,0.2,200,0.8,0.01,0.20,0.3,1500,0.8,0.01,0.20,0.3,1500,"
            "0.8,0.01,0.20,0.4,1500,0.8,0.01,0.20,0.5,900,2.0,30.0,5.0,.3666667,.4736842,1,1 "
            "--source=SDR,22.5,90,0 --source-loc=0,0,-10 --frequency=2.0 --timetolive=350 "
            f"--binsize=10.0 --toa-degree={deg} --overridemfp=25,50 --nodeflect "
            "--seis-p2p=0,67.5,0,950,67.5,0,1.0,2.0,40.0,16 "
            "--seis-p2p=0,112.5,0,950,112.5,0,1.0,2.0,40.0,16 "
            "--seis-p2p=0,90,0,950,90,0,1.0,2.0,40.0,16").split()


# Dense scatter-event grid of BASELINE config 5: 2 wave types x 300 frames
# x 64 x 256 x 256 uint32 = 10 GB over the crust-pinch model's footprint and the run's 350 s.
CRUSTPINCH_VOLUME = dict(origin=(-1000.0, -1000.0, -250.0), cell_size=(2000.0 / 256, 2000.0 / 256, 250.0 / 64),
                         dims=(256, 256, 64), n_frames=300, frame_dt=350.0 / 300)


def toysphere_vids(deg=4):
    """do-toysphere-vids.sh:21-60: model 30, pinned mean free paths, scattering without
    deflection, raw output coordinates (a ray-path video run)."""
    return ("--grid-compiled=30 --model-args=0.8,0.01,0.5,0.2,50,0.8,0.01,0.5,0.3,1000,0.8,0.01,0.7,0.5,300 "
            "--source=SDR,22.5,90,0 --source-loc=0,0,-10 --frequency=2.0 --timetolive=6000 "
            f"--binsize=60.0 --toa-degree={deg} --overridemfp=50,40 --nodeflect --ocsraw "
            "--seis-p2p=0,0,0,1000,0,0,1.0,2.0,40.0,16 --seis-p2p=0,90,0,1000,90,0,1.0,2.0,40.0,16").split()


def lopnor_vids(deg=4):
    """do-lopnor-vids.sh:21-100 with event=eq: model 21 (Moho transition, gradual profile),
    Earth-flattened, four scattering regions, pinned mean free paths, no deflection."""
    return ("--grid-compiled=21 --flatten --range=1200 "
            "--model-args=0.8,0.01,0.5,0.2,50,0.8,0.01,0.5,0.3,1000,0.8,0.01,0.7,0.5,300,0.8,0.02,0.5,0.5,2000 "
            "--source=SDR,125,40,90,0.0 --source-loc=425.54,-169.53,-31.02 --frequency=2.0 --timetolive=350 "
            f"--binsize=10.0 --toa-degree={deg} --overridemfp=1,1 --nodeflect "
            "--seis-p2p=425.54,-169.53,0.98,-390.04,-167.18,1.457,1.0,2.0,40.0,16 "
            "--seis-p2p=425.54,-169.53,0.98,-102.27,430.84,0.60,1.0,2.0,40.0,16").split()


def upthrust(deg=4):
    """Model 8 (crust upthrust, tetra) with the crust-pinch run's source and arrays
    (no reference run script uses it; user_Upthrust_inc.cpp argument pattern of 32)."""
    return ("--grid-compiled=8 "
            "--model-args=0.8,0.01,0.20,0.2,200,0.8,0.01,0.20,0.3,1500,0.8,0.02,0.30,0.3,1200,"
            "0.8,0.01,0.20,0.4,1500,0.8,0.01,0.20,0.5,900,2.0,30.0,10.0,-12,-3,1.5,4 "
            "--source=SDR,22.5,90,0 --source-loc=0,0,-10 --frequency=2.0 --timetolive=600 "
            f"--binsize=2.00 --toa-degree={deg} "
            "--seis-p2p=0,67.5,0,950,67.5,0,1.0,2.0,40.0,40 "
            "--seis-p2p=0,112.5,0,950,112.5,0,1.0,2.0,40.0,40").split()


def lopnor_moho(deg=4, selector=1):
    """The Lop Nor model with its Moho transition layers: selectors 1-4 of the reference's dispatcher given 20 or more
    model arguments (user.cpp:69-80 -> LopNorCylinderMoho, user_LopNorCylMoho_inc.cpp; four scattering regions of five
    numbers each), run as do-lopnor.sh runs the baseline model (earthquake source, the script's two arrays)."""
    return (f"--grid-compiled={selector} --flatten --range=1200 "
            "--model-args=0.8,0.06,0.25,0.5,250,0.8,0.05,0.5,0.5,1000,0.8,0.04,1.0,0.5,2000,0.7,0.03,0.4,0.3,1500 "
            "--source=SDR,125,40,90,0.0 --source-loc=425.54,-169.53,-31.02 --frequency=2.0 --timetolive=600 "
            f"--binsize=2.00 --toa-degree={deg} "
            "--seis-p2p=425.54,-169.53,0.98,-390.04,-167.18,1.457,1.0,2.0,40.0,40 "
            "--seis-p2p=425.54,-169.53,0.98,-102.27,430.84,0.60,1.0,2.0,40.0,40").split()


def scat_params_study(deg=4):
    """Selector 128 (user.cpp:115-118 -> ScatParamsStudy, user.cpp:200-310: a stack of layers that differ in one
    scattering parameter at a time), with the half-space run's source and arrays (no reference run script uses it)."""
    return ("--grid-compiled=128 --range=300 --source=SDR,0,90,0,0.0 --source-loc=0,0,-5 --frequency=2.0 "
            f"--timetolive=200 --binsize=0.50 --toa-degree={deg} "
            "--seis-p2p=0,0,0,100,100,0,2.737,0.105,10.0,24 --seis-p2p=0,0,0,140,0,0,2.737,0.105,10.0,24").split()


def halfspace_one(deg=9):
    """BASELINE config 1 as it names it: the half-space run with one receiver."""
    return halfspace(deg, one_receiver=True)


CONFIGS = {"halfspace": halfspace, "halfspace_one": halfspace_one, "crustpinch": crustpinch, "crustpinch_vids": crustpinch_vids, "lopnor": lopnor, "sphere": sphere, "sphere_deep": sphere_deep,
           "toysphere_vids": toysphere_vids, "lopnor_vids": lopnor_vids, "upthrust": upthrust,
           "lopnor_moho": lopnor_moho, "lopnor_moho_sel3": lambda deg=4: lopnor_moho(deg, selector=3),
           "scat_params_study": scat_params_study}
