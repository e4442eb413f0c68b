from .tangent_nonlin import TangentNonLin
from .tangent_lin import TangentLin
from .tangent_perceptron import TangentPerceptron
from .trans_field import TransField
from .field_conv import FieldConv
from .echo import ECHO
from .lift_block import LiftBlock
from .fc_resnet_block import FCResNetBlock
from .echo_block import ECHOBlock

__all__ = ['TangentNonLin', 'TangentLin', 'TangentPerceptron', 'TransField', 'FieldConv', 'ECHO', 'LiftBlock',
           'FCResNetBlock', 'ECHOBlock']
