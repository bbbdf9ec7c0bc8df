"""Host-side calendar logic of the threshold() path: get_calendar() and
add_doy() of the reference (xmhw/identify.py:82-134, :28-79) on numpy
datetime64 (or cftime-like objects exposing year/month/dayofyr).

The doy array is integer index work: it must be bit-exact.
"""
import numpy as np

from .exception import XmhwException

_NDAYS = {  # identify.py:104-113
    "standard": 365.25, "gregorian": 365.25, "proleptic_gregorian": 365.25,
    "all_leap": 366, "noleap": 365, "365_day": 365, "360_day": 360, "julian": 365.25,
}


def get_calendar(calendar):
    """Days per year for a calendar name (identify.py:120-133).  '' / unknown
    -> 365.25 like the reference (which prints a note and carries on)."""
    if calendar is None:
        calendar = ""
    if calendar in ["360", "365", "366"]:
        calendar = f"{calendar}_day"
    elif calendar == "leap":
        calendar = "standard"
    if calendar not in _NDAYS:
        return 365.25
    return _NDAYS[calendar]


def calendar_of(time_values, encoding=None, attrs=None):
    """identify.py:114-119: encoding, then attrs, then the first value's
    ``calendar`` attribute (cftime objects)."""
    if encoding and "calendar" in encoding:
        return encoding["calendar"]
    if attrs and "calendar" in attrs:
        return attrs["calendar"]
    tv = np.asarray(time_values)
    first = tv.flat[0] if tv.size else None
    return getattr(first, "calendar", "")


def _fields(time_values):
    """year, month, dayofyear, is_leap_year of a time axis."""
    tv = np.asarray(time_values)
    if tv.dtype.kind == "M":
        t = tv.astype("datetime64[D]")
        years = t.astype("datetime64[Y]")
        year = years.astype(np.int64) + 1970
        month = (t.astype("datetime64[M]").astype(np.int64) % 12) + 1
        dayofyear = (t - years.astype("datetime64[D]")).astype(np.int64) + 1
        leap = ((year % 4 == 0) & (year % 100 != 0)) | (year % 400 == 0)
        return year, month, dayofyear, leap
    if tv.dtype == object and tv.size and hasattr(tv.flat[0], "year"):
        year = np.array([t.year for t in tv.flat], dtype=np.int64)
        month = np.array([t.month for t in tv.flat], dtype=np.int64)
        dayofyear = np.array([getattr(t, "dayofyr", None) or t.timetuple().tm_yday
                              for t in tv.flat], dtype=np.int64)
        cal = getattr(tv.flat[0], "calendar", "standard")
        if cal in ("noleap", "365_day", "360_day"):
            leap = np.zeros(year.shape, bool)
        elif cal in ("all_leap", "366_day"):
            leap = np.ones(year.shape, bool)
        elif cal == "julian":
            leap = year % 4 == 0
        else:
            leap = ((year % 4 == 0) & (year % 100 != 0)) | (year % 400 == 0)
        return year, month, dayofyear, leap
    raise XmhwException("time axis must be datetime64 or cftime-like objects")


def years_of(time_values):
    return _fields(time_values)[0]


def add_doy(time_values, keep_tstep=False):
    """int64 doy label per time step (identify.py:28-79).

    daily: doy = dayofyear + (not leap and month >= 3) -> 366-slot calendar
    (Feb 29 = 60, skipped in non-leap years).  keep_tstep: the number of steps
    in the SECOND year defines the cycle; the series must be a whole number of
    cycles.
    """
    year, month, dayofyear, leap = _fields(time_values)
    if keep_tstep is True:
        years = np.unique(year)
        if years.size < 2:
            # the reference indexes years[1] and fails with IndexError
            raise XmhwException("To use original timestep as climatology base unit, "
                                "timeseries has to have complete years")
        n = int(np.sum(year == years[1]))
        if len(year) % n != 0:
            raise XmhwException("To use original timestep as climatology base unit, "
                                "timeseries has to have complete years")
        return np.tile(np.arange(1, n + 1, dtype=np.int64), len(year) // n)
    return (dayofyear + ((~leap) & (month >= 3))).astype(np.int64)
