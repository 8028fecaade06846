"""TEST INFRASTRUCTURE (see oracle/__init__.py): CPU restatement of the survey-tile sampler and of the TAN-SIP world
coordinates it reports.  PARITY UNPINNED for the FITS / WCS part: astropy (``fits.open``, ``WCS.all_pix2world``), which the
reference calls at utils/dataloaders.py:417-433, is absent here, so these follow the published algorithms (FITS standard 4.0;
Calabretta & Greisen 2002; Shupe et al. 2005) and are cross-checked only against an independent formulation below.

* ``cutouts_np``: utils/dataloaders.py:449-476 + :618-621 for given window corners (crop, then clip with NaN kept).
* ``tan_sip_pix2world``: pixel -> (RA, Dec) through explicit 3-D rotations of the native sphere -- a different route from
  ``fits_lite.TanSipWCS``'s closed-form standard-coordinate inversion.
"""
import numpy as np


def cutouts_np(tile, hs, ws, S, pixel_min=None, pixel_max=None):
    out = np.stack([tile[:, h:h + S, w:w + S] for h, w in zip(hs, ws)]).astype(np.float32)
    if pixel_min is not None:
        out[out < pixel_min] = pixel_min          # NaN < x is False: NaNs stay (dataloaders.py:618-619)
    if pixel_max is not None:
        out[out > pixel_max] = pixel_max
    return out


def tan_sip_pix2world(header, x, y, origin=0):
    """(ra, dec) [deg] of pixel (x = FITS axis 1, y = axis 2).  Native sphere of the gnomonic projection: pole at the
    reference point, R_theta = (180/pi) cot(theta); rotated to the celestial sphere with Euler angles
    (alpha_p, delta_p, phi_p) = (CRVAL1, CRVAL2, 180 deg)."""
    x = np.asarray(x, dtype=np.float64) + (1 - origin)
    y = np.asarray(y, dtype=np.float64) + (1 - origin)
    u, v = x - float(header["CRPIX1"]), y - float(header["CRPIX2"])
    if str(header.get("CTYPE1", "")).endswith("-SIP"):
        def poly(name):
            order = int(header.get(f"{name}_ORDER", 0))
            tot = np.zeros_like(u)
            for p in range(order + 1):
                for q in range(order + 1 - p):
                    c = header.get(f"{name}_{p}_{q}")
                    if c is not None:
                        tot = tot + float(c) * u ** p * v ** q
            return tot
        u, v = u + poly("A"), v + poly("B")
    cd = np.array([[float(header.get("CD1_1", 0.0)), float(header.get("CD1_2", 0.0))],
                   [float(header.get("CD2_1", 0.0)), float(header.get("CD2_2", 0.0))]])
    px = cd[0, 0] * u + cd[0, 1] * v          # intermediate world coordinates, degrees
    py = cd[1, 0] * u + cd[1, 1] * v
    # native spherical coordinates (paper II eqs. 14, 15, 55): phi = arg(-y, x), R = hypot, theta = atan(180 / (pi R))
    r = np.hypot(px, py)
    phi = np.arctan2(px, -py)
    theta = np.arctan2(180.0 / np.pi, r)
    # unit vector on the native sphere, then rotate: native pole -> (alpha_p, delta_p), native longitude of the celestial pole 180 deg
    nx, ny, nz = np.cos(theta) * np.cos(phi), np.cos(theta) * np.sin(phi), np.sin(theta)
    ap, dp, pp = np.deg2rad(float(header["CRVAL1"])), np.deg2rad(float(header["CRVAL2"])), np.pi
    # inverse of the celestial -> native rotation R = Rz(phi_p) Rx? expressed with paper II eq. 2
    sin_d = np.sin(theta) * np.sin(dp) + np.cos(theta) * np.cos(dp) * np.cos(phi - pp)
    dec = np.arcsin(np.clip(sin_d, -1.0, 1.0))
    ra = ap + np.arctan2(-np.cos(theta) * np.sin(phi - pp), np.sin(theta) * np.cos(dp) - np.cos(theta) * np.sin(dp) * np.cos(phi - pp))
    del nx, ny, nz
    return np.mod(np.rad2deg(ra), 360.0), np.rad2deg(dec)
