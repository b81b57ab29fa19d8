"""Import shims named like the reference's ``pyfiles/`` modules.

The train notebooks start with ``sys.path.append("../pyfiles/")`` followed by ``from util import ...``, ``from dataset import
...``, ``from model import ...``, ``from util_notebook import ...`` (05-train cell 1).  Pointing that ``sys.path`` entry at this
directory instead makes the same import lines resolve to the MI355X implementations; nothing else in the cell changes."""
