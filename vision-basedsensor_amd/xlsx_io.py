"""Minimal `.xlsx` reader / writer on the standard library (zipfile + xml.etree), for the three tables the hot path
exchanges as Excel files: the parameter sheets `IntrinsicParameters.xlsx` (`intrinsic_calibration.py:33-51`) and
`ExtrinsicParameters.xlsx` (`extrinsic_calibration.py:125-151`) that `MarkerAnalysis.load_parameters` reads
(`3d_reconstruction.py:83-124`), and the result sheet `marker_3d_coordinates.xlsx` it writes (`:430-433`).

The reference goes through pandas + openpyxl; neither Excel engine is needed here.  Scope: the first worksheet,
one header row, cells that are numbers, shared / inline / formula strings or booleans — what `DataFrame.to_excel`
produces.  Styles, dates-as-numbers formatting, merged cells and formulas' cached errors are ignored.
"""
from __future__ import annotations

import re
import zipfile
from pathlib import Path
from typing import List, Sequence
from xml.etree import ElementTree as ET
from xml.sax.saxutils import escape

_NS = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main",
       "r": "http://schemas.openxmlformats.org/officeDocument/2006/relationships",
       "p": "http://schemas.openxmlformats.org/package/2006/relationships"}


def _col_index(ref: str) -> int:
    """'A1' -> 0, 'AB7' -> 27."""
    n = 0
    for ch in ref:
        if not ch.isalpha():
            break
        n = n * 26 + (ord(ch.upper()) - 64)
    return n - 1


def _col_name(i: int) -> str:
    s = ""
    i += 1
    while i:
        i, r = divmod(i - 1, 26)
        s = chr(65 + r) + s
    return s


def _text(si) -> str:
    """A shared-string / inline-string item: plain <t> or rich-text runs <r><t>."""
    return "".join(t.text or "" for t in si.iter("{%s}t" % _NS["m"]))


def _first_sheet_path(z: zipfile.ZipFile) -> str:
    names = set(z.namelist())
    try:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        sheet = wb.find("m:sheets/m:sheet", _NS)
        rid = sheet.get("{%s}id" % _NS["r"])
        rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
        for rel in rels.findall("p:Relationship", _NS):
            if rel.get("Id") == rid:
                target = rel.get("Target").lstrip("/")
                path = target if target.startswith("xl/") else "xl/" + target
                if path in names:
                    return path
    except (KeyError, AttributeError, ET.ParseError):
        pass
    cands = sorted(n for n in names if re.fullmatch(r"xl/worksheets/sheet\d+\.xml", n))
    if not cands:
        raise ValueError("no worksheet in workbook")
    return cands[0]


def read_rows(path) -> List[list]:
    """All rows of the first worksheet as lists (None for empty cells), ragged rows padded to the widest."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"No such file: {path}")
    try:
        z = zipfile.ZipFile(path)
    except zipfile.BadZipFile as exc:
        raise ValueError(f"{path} is not an .xlsx workbook: {exc}") from None
    with z:
        shared: List[str] = []
        if "xl/sharedStrings.xml" in z.namelist():
            sst = ET.fromstring(z.read("xl/sharedStrings.xml"))
            shared = [_text(si) for si in sst.findall("m:si", _NS)]
        sheet = ET.fromstring(z.read(_first_sheet_path(z)))
    rows, width = [], 0
    for row in sheet.iterfind("m:sheetData/m:row", _NS):
        out, nxt = [], 0
        for c in row.findall("m:c", _NS):
            ref = c.get("r")
            ci = _col_index(ref) if ref else nxt
            nxt = ci + 1
            while len(out) < ci:
                out.append(None)
            t = c.get("t", "n")
            v = c.find("m:v", _NS)
            if t == "inlineStr":
                is_ = c.find("m:is", _NS)
                val = _text(is_) if is_ is not None else ""
            elif v is None or v.text is None:
                val = None
            elif t == "s":
                val = shared[int(v.text)]
            elif t in ("str", "e"):
                val = v.text
            elif t == "b":
                val = v.text.strip() not in ("0", "")
            else:
                f = float(v.text)
                val = int(f) if (f.is_integer() and "." not in v.text and "e" not in v.text.lower()) else f
            out.append(val)
        # keep the sheet's own row numbering (blank rows are real rows in the reference's extrinsics sheet)
        rn = row.get("r")
        if rn is not None:
            while len(rows) < int(rn) - 1:
                rows.append([])
        rows.append(out)
        width = max(width, len(out))
    return [r + [None] * (width - len(r)) for r in rows]


def read_xlsx(path):
    """First worksheet -> pandas.DataFrame, first row = header (what `pd.read_excel(path)` returns for such files)."""
    import pandas as pd
    rows = read_rows(path)
    if not rows:
        return pd.DataFrame()
    header = [("Unnamed: %d" % i) if h is None else str(h) for i, h in enumerate(rows[0])]
    return pd.DataFrame(rows[1:], columns=header)


def _cell(ref: str, v) -> str:
    import math
    import numbers
    if v is None:
        return ""
    if isinstance(v, bool):
        return f'<c r="{ref}" t="b"><v>{int(v)}</v></c>'
    if isinstance(v, numbers.Integral):
        return f'<c r="{ref}"><v>{int(v)}</v></c>'
    if isinstance(v, numbers.Real):
        fv = float(v)
        if math.isnan(fv) or math.isinf(fv):
            return ""                                   # Excel has no NaN / inf: empty cell, like to_excel
        return f'<c r="{ref}"><v>{repr(fv)}</v></c>'    # repr round-trips the float64 exactly
    s = escape(str(v))
    keep = ' xml:space="preserve"' if s != s.strip() else ""
    return f'<c r="{ref}" t="inlineStr"><is><t{keep}>{s}</t></is></c>'


def write_xlsx(path, columns: Sequence[str], rows, sheet_name: str = "Sheet1") -> None:
    """One worksheet: a header row of `columns`, then `rows` (iterable of sequences).  Numbers are stored as
    numbers with their shortest round-trip decimal, everything else as inline strings."""
    lines = []
    hdr = "".join(_cell(f"{_col_name(i)}1", str(c)) for i, c in enumerate(columns))
    lines.append(f'<row r="1">{hdr}</row>')
    n = 1
    for row in rows:
        n += 1
        cells = "".join(_cell(f"{_col_name(i)}{n}", v.item() if hasattr(v, "item") else v) for i, v in enumerate(row))
        lines.append(f'<row r="{n}">{cells}</row>')
    last = f"{_col_name(max(len(columns), 1) - 1)}{n}"
    sheet = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
             f'<worksheet xmlns="{_NS["m"]}"><dimension ref="A1:{last}"/><sheetData>' + "".join(lines) +
             "</sheetData></worksheet>")
    content_types = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
                     '<Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
                     '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>'
                     '<Default Extension="xml" ContentType="application/xml"/>'
                     '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.'
                     'spreadsheetml.sheet.main+xml"/>'
                     '<Override PartName="/xl/worksheets/sheet1.xml" ContentType="application/vnd.openxmlformats-'
                     'officedocument.spreadsheetml.worksheet+xml"/></Types>')
    rels = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
            f'<Relationships xmlns="{_NS["p"]}"><Relationship Id="rId1" Type="http://schemas.openxmlformats.org/'
            'officeDocument/2006/relationships/officeDocument" Target="xl/workbook.xml"/></Relationships>')
    workbook = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
                f'<workbook xmlns="{_NS["m"]}" xmlns:r="{_NS["r"]}"><sheets>'
                f'<sheet name="{escape(sheet_name)}" sheetId="1" r:id="rId1"/></sheets></workbook>')
    wb_rels = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
               f'<Relationships xmlns="{_NS["p"]}"><Relationship Id="rId1" Type="http://schemas.openxmlformats.org/'
               'officeDocument/2006/relationships/worksheet" Target="worksheets/sheet1.xml"/></Relationships>')
    path = Path(path)
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("[Content_Types].xml", content_types)
        z.writestr("_rels/.rels", rels)
        z.writestr("xl/workbook.xml", workbook)
        z.writestr("xl/_rels/workbook.xml.rels", wb_rels)
        z.writestr("xl/worksheets/sheet1.xml", sheet)


def dataframe_to_xlsx(df, path, sheet_name: str = "Sheet1") -> None:
    """`df.to_excel(path, index=False)` without an Excel engine."""
    write_xlsx(path, [str(c) for c in df.columns], df.itertuples(index=False, name=None), sheet_name)
