"""Drop-in mirror of the reference's ``dust3r`` package for the hot path only
(model / inference / image_pairs / cloud_opt).  ``align3r_amd.install_as_dust3r()`` registers these
modules under the reference's own import names so that tool/*.py-style drivers run unchanged."""
