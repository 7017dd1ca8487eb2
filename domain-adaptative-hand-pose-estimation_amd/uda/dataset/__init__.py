from .rendered_hand_pose import RenderedHandPose
from .STB import STB, STBx1
from .hand_3d_studio import Hand3DStudio, Hand3DStudioAll
