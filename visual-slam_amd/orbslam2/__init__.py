"""Drop-in `orbslam2` package for the hot path of p2004dr/visual-slam: same module names, class names, constructor
and method signatures as the reference's src/orbslam2/{extractor,matcher,initializer,utils}.py, with every cv2
call on the path replaced by the MI355X HIP library (vslam_amd / include/vslam_amd.h).  The reference's Tracker
and LocalMapper stay the caller."""
__version__ = "0.1.0"
